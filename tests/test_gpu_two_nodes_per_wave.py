"""(Also covers DEFERRED classification -- step_pipe.cuh: CIN -- which forwards of this size with the fp32 edge state use: a message step's
classified state is classified by the step that reads it back; the comparison batches below are small enough to classify in place.)
The two-nodes-per-wave form of the buffer-addressed step kernel (step_pipe.cuh: NPW = 2; message steps that do not classify, batches
of >= 16 384 nodes whose nodes average <= 128 edges).  A wave computes its second node with the arithmetic and in the order a wave
of its own would: with `encoder_unsplit` (a node's encoder output independent of the batch around it) the logits of a graph inside such
a batch must be BIT FOR BIT those of the same graph inside a batch small enough for one node per wave -- and within rounding of the CPU
oracle on the graph alone.  The batches are built to reach every hand-over: one round -> one round (the hooked path), a lone chunk,
an isolated first / second node (cold start), a first node with several rounds, an odd node count, unsorted rows."""
import numpy as np
import pytest
import torch

from oracle.mpn_oracle import NumpyOracle
from test_gpu_parity import Data, _default_model, _dense_graph, build

pytestmark = pytest.mark.gpu


def _ragged_graph(n, rng, drop=0.0, isolate=()):
    """dense graph on n nodes minus a random share of edges; nodes in `isolate` lose every out-edge (empty segments)."""
    ei = _dense_graph(n)
    keep = rng.random(ei.shape[1]) >= drop
    for v in isolate:
        keep &= ei[0] != v
    return ei[:, keep]


def _batch(sizes, rng, drop=0.0, isolate_every=0):
    parts, off, ptr = [], 0, [0]
    for g, n in enumerate(sizes):
        iso = ()
        if isolate_every and g % isolate_every == 0:
            iso = (0, 1, 4) if g % (2 * isolate_every) == 0 else (1, n - 1)      # even and odd positions: first and second node of a wave
        ei = _ragged_graph(n, rng, drop, iso)
        parts.append(ei + off)
        off += n
        ptr.append(ptr[-1] + ei.shape[1])
    return np.concatenate(parts, axis=1), off, ptr


def _inputs(n, e, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    return x, rng.random((e, 4)).astype(np.float32)


def _run(m, x, ei, ea):
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
    assert m.graph_flags() & 2 == 0      # (4 = IRREGULAR is informational: a degree above the padded stride, compact layout used)
    return out


CASES = {
    # name: (graph sizes, edge drop share, isolate every k-th graph, model overrides, module options)
    "dense128_pairs": ([128] * 136, 0.0, 0, {}, {}),                                          # N = 17 408: one round -> one round everywhere
    "ragged_100_128": (None, 0.1, 5, {}, {}),                                                  # sizes drawn below; isolated first / second nodes
    "lone_chunks": ([60] * 300, 0.2, 7, {}, {}),                                               # <= 59 edges per node: compute1 + the hook fired up front
    "a_few_big_nodes": ([300] * 3 + [90] * 190, 0.0, 0, {}, {}),                               # nodes with 3 rounds between one-round nodes; odd N below
    "six_steps_sum_fp32": ([120] * 140, 0.05, 9, {"num_enc_steps": 6}, {}),                    # deferred classification over steps 4, 5, 6
    "one_class_step": ([120] * 140, 0.0, 0, {"num_class_steps": 1}, {}),                       # only the last step classifies: nothing to defer
    "six_steps_mean_bf16": ([120] * 140, 0.05, 9, {"num_enc_steps": 6, "node_agg_fn": "mean"}, {"edge_state_dtype": "bf16"}),   # non-first non-classifying steps
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_two_nodes_per_wave_is_bitwise_one_node_per_wave_and_matches_the_oracle(name):
    sizes, drop, iso, over, opts = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    if sizes is None:
        sizes = [int(v) for v in rng.integers(100, 129, size=160)]
    if name == "a_few_big_nodes":
        sizes = sizes + [77]                                                                   # odd total: the last wave has no second node
    ei, n, ptr = _batch(sizes, rng, drop, iso)
    assert n >= 16384 and ei.shape[1] / n <= 128
    scale = 1.0 if over.get("node_agg_fn") == "mean" else 1.0 / 127
    params, arch, sd = _default_model(scale, **over)
    m = build(params, arch, sd)
    m.encoder_unsplit = True            # a node's encoder output does not depend on the batch around it (and one wave per node is pinned)
    for k, v in opts.items():
        setattr(m, k, v)
    x, ea = _inputs(n, ei.shape[1], 3)
    whole = _run(m, x, ei, ea)
    # the same graphs in three batches of < 16 384 nodes each: one node per wave
    cuts = [0, len(sizes) // 3, 2 * len(sizes) // 3, len(sizes)]
    node_ptr = np.concatenate([[0], np.cumsum(sizes)])
    for a, b in zip(cuts[:-1], cuts[1:]):
        n0, n1, e0, e1 = node_ptr[a], node_ptr[b], ptr[a], ptr[b]
        assert 4096 <= n1 - n0 < 16384
        part = _run(m, x[n0:n1], ei[:, e0:e1] - n0, ea[e0:e1])
        for w, q in zip(whole, part):
            assert torch.equal(w[e0:e1], q), float((w[e0:e1] - q).abs().max())
    # and the oracle on single graphs of the batch (the first, one with isolated nodes, the last)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tol = 2e-4 if opts.get("edge_state_dtype") == "bf16" else 2e-5
    for g in {0, (iso or 1), len(sizes) - 1}:
        n0, n1, e0, e1 = node_ptr[g], node_ptr[g + 1], ptr[g], ptr[g + 1]
        ref = orc.forward(x[n0:n1], ei[:, e0:e1] - n0, ea[e0:e1])
        for w, r in zip(whole, ref):
            assert np.abs(w[e0:e1].cpu().numpy() - r).max() <= tol


def test_unsorted_rows_in_a_two_nodes_per_wave_batch():
    """Shuffled edge order: the device sort repairs the plan, every step reads the permutation -- the hand-over must carry the second
    node's permutation entries as well.  Against the sorted run of the same batch (same sums in the caller's order per segment: the
    stable sort keeps it), within rounding, and against the oracle on one graph."""
    rng = np.random.default_rng(5)
    sizes = [110] * 150
    ei, n, ptr = _batch(sizes, rng, 0.0, 0)
    params, arch, sd = _default_model(1.0 / 109)
    m = build(params, arch, sd)
    x, ea = _inputs(n, ei.shape[1], 4)
    ref = _run(m, x, ei, ea)
    perm = rng.permutation(ei.shape[1])
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei[:, perm]).cuda(), torch.from_numpy(ea[perm]).cuda())
    with torch.no_grad():
        got = [t.clone() for t in m(d)["classified_edges"]]
    assert m.graph_flags() == 1
    for g, r in zip(got, ref):
        assert float((g.cpu() - r.cpu()[perm]).abs().max()) <= 2e-5


def test_bad_index_poisons_every_logit_slot_with_deferred_classification():
    """An out-of-range node id: every step returns early and every classified step's slot is NaN -- also the slots that the deferred
    scheme fills from the NEXT step's launch."""
    rng = np.random.default_rng(9)
    ei, n, _ = _batch([128] * 130, rng)
    ei = ei.copy()
    ei[1, 12345] = n + 7
    params, arch, sd = _default_model(1.0 / 127)
    m = build(params, arch, sd)
    x, ea = _inputs(n, ei.shape[1], 2)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    with torch.no_grad():
        out = m(d)["classified_edges"]
    assert m.graph_flags() & 2
    assert len(out) == 3 and all(bool(torch.isnan(t).all()) for t in out)

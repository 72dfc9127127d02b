"""Column ranges (round 4; an option, off by default -- measured: no gain): steps 2 ... L of the specialised step kernels compute the target ids of a segment from the per-node
(start1, len1, start2) step 1 derives, when every node's target ids are <= 2 contiguous runs -- the shape of every graph the reference
builds (inference.py:209-216) and of dense graphs; any other forward streams col32 on every step.  Pinned here: the verdict
(`column_ranges_state`) for each graph family, and logits BIT FOR BIT equal to the streaming path (`model.column_ranges = False`,
the default; True = GNNCCA_OPT_COLUMN_RANGES) -- on the f32 kernel of graphs up to 512 nodes and on the buffer-addressed kernel beyond."""
import numpy as np
import pytest
import torch

from oracle.mpn_oracle import NumpyOracle
from test_gpu_parity import Data, _default_model, _dense_graph, build

pytestmark = pytest.mark.gpu


def cross_camera_graph(cam_sizes, offset=0):
    """inference.py:209-216: cameras in order, nodes ascending inside a camera, every node to every node of the OTHER cameras."""
    n = int(sum(cam_sizes))
    cam = np.repeat(np.arange(len(cam_sizes)), cam_sizes)
    rows, cols = [], []
    for i in range(n):
        others = np.nonzero(cam != cam[i])[0]
        rows.append(np.full(len(others), i))
        cols.append(others)
    return np.stack([np.concatenate(rows), np.concatenate(cols)]).astype(np.int64) + offset, n


def union(parts):
    eis, off = [], 0
    for make in parts:
        ei, n = make(off)
        eis.append(ei)
        off += n
    return np.concatenate(eis, axis=1), off


def run_both(params, arch, sd, x, ei, ea, **opts):
    m = build(params, arch, sd)
    for k, v in opts.items():
        setattr(m, k, v)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    with torch.no_grad():
        m.column_ranges = True
        a = [t.clone() for t in m(d)["classified_edges"]]
        state = m.column_ranges_state()
        flags = m.graph_flags()
        m.column_ranges = False
        b = [t.clone() for t in m(d)["classified_edges"]]
        assert m.column_ranges_state() == 0      # never asked
    return a, b, state, flags


def inputs(n, e, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    return x, rng.random((e, 4)).astype(np.float32)


CASES = {
    # name: (graph builder -> (edge_index, n), expected verdict 0 = ranges used)
    "dense256": (lambda: (_dense_graph(256), 256), 0),
    "dense40": (lambda: (_dense_graph(40), 40), 0),
    "terrace_4x8": (lambda: cross_camera_graph([8, 8, 8, 8]), 0),
    "ragged_cams": (lambda: cross_camera_graph([3, 0, 11, 1, 7]), 0),
    "batch_64x_dense128": (lambda: union([lambda o: (_dense_graph(128, o), 128)] * 64), 0),          # N = 8192: buffer-addressed kernel
    "batch_mixed": (lambda: union([lambda o: cross_camera_graph([5, 9, 4, 12], o), lambda o: (_dense_graph(300, o), 300),
                                   lambda o: cross_camera_graph([40, 25, 61], o), lambda o: (_dense_graph(77, o), 77)] * 3), 0),
    "dense900": (lambda: (_dense_graph(900), 900), 0),                                                 # LDS gather-table variant
    "dense1500": (lambda: (_dense_graph(1500), 1500), 0),                                              # four waves per node
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_two_run_graphs_use_the_ranges_and_give_the_same_bits(name):
    make, want_state = CASES[name]
    ei, n = make()
    deg = np.bincount(ei[0], minlength=n)
    params, arch, sd = _default_model(1.0 / max(int(deg.max()), 1))
    x, ea = inputs(n, ei.shape[1], 5)
    a, b, state, flags = run_both(params, arch, sd, x, ei, ea)
    assert flags == 0 and state == want_state
    for s, t in zip(a, b):
        assert torch.equal(s, t), float((s - t).abs().max())
    if n <= 300:   # and the oracle, for the small ones
        ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
        for s, r in zip(a, ref):
            assert np.abs(s.cpu().numpy() - r).max() <= 2e-5


def _three_runs(n, offset=0):
    """dense graph minus the edge i -> i + 2 (mod n): most nodes then have THREE runs of target ids."""
    ei = _dense_graph(n)
    keep = ei[1] != (ei[0] + 2) % n
    return ei[:, keep] + offset, n


@pytest.mark.parametrize("kind", ["three_runs_small", "three_runs_batch", "one_bad_node_in_a_batch", "shuffled", "descending", "duplicates"])
def test_other_graphs_stream_the_ids_and_give_the_same_bits(kind):
    rng = np.random.default_rng(17)
    if kind == "three_runs_small":
        ei, n = _three_runs(90)
    elif kind == "three_runs_batch":
        ei, n = union([lambda o: _three_runs(100, o)] * 12)
    elif kind == "one_bad_node_in_a_batch":     # 1199 good graphs' worth of nodes, ONE node with three runs: the whole forward streams
        ei, n = union([lambda o: (_dense_graph(120, o), 120)] * 10)
        bad = (ei[0] == 601) & (ei[1] == 660)
        ei = ei[:, ~bad]
    elif kind == "shuffled":                     # unsorted rows: after the stable device sort a segment keeps the caller's (random) order
        ei = _dense_graph(70)
        ei = ei[:, rng.permutation(ei.shape[1])]
        n = 70
    elif kind == "descending":                   # sorted rows, targets descending: every edge is a "break"
        ei = _dense_graph(64)
        order = np.lexsort((-ei[1], ei[0]))
        ei, n = ei[:, order], 64
    else:                                        # duplicate edges: a repeated target id is a break as well
        ei = _dense_graph(50)
        ei = np.concatenate([ei, ei[:, ::7]], axis=1)
        ei, n = ei[:, np.argsort(ei[0], kind="stable")], 50
    deg = np.bincount(ei[0], minlength=n)
    params, arch, sd = _default_model(1.0 / max(int(deg.max()), 1))
    x, ea = inputs(n, ei.shape[1], 6)
    a, b, state, flags = run_both(params, arch, sd, x, ei, ea)
    assert state == 1
    assert flags == (1 if kind == "shuffled" else 0)
    for s, t in zip(a, b):
        assert torch.equal(s, t), float((s - t).abs().max())
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    for s, r in zip(a, ref):
        assert np.abs(s.cpu().numpy() - r).max() <= 2e-5


def test_verdict_is_per_forward_and_shapes_may_alternate():
    """The ranges and their verdict live in the forward's workspace: a two-run graph after a three-run graph (and back) on ONE module,
    same stream, same workspace -- nothing of the previous forward may leak (stale ranges, a stale verdict)."""
    params, arch, sd = _default_model(1.0 / 99)
    m = build(params, arch, sd)
    m.column_ranges = True
    m_ref = build(params, arch, sd)
    m_ref.column_ranges = False
    good, n_g = cross_camera_graph([30, 20, 50])
    bad, n_b = _three_runs(100)
    xg, eag = inputs(n_g, good.shape[1], 1)
    xb, eab = inputs(n_b, bad.shape[1], 2)
    dg = Data(torch.from_numpy(xg).cuda(), torch.from_numpy(good).cuda(), torch.from_numpy(eag).cuda())
    db = Data(torch.from_numpy(xb).cuda(), torch.from_numpy(bad).cuda(), torch.from_numpy(eab).cuda())
    with torch.no_grad():
        want_g = [t.clone() for t in m_ref(dg)["classified_edges"]]
        want_b = [t.clone() for t in m_ref(db)["classified_edges"]]
        for d, want, state in ((dg, want_g, 0), (db, want_b, 1), (dg, want_g, 0), (dg, want_g, 0), (db, want_b, 1)):
            got = m(d)["classified_edges"]
            assert m.column_ranges_state() == state
            for s, t in zip(got, want):
                assert torch.equal(s, t)


def test_bf16_edge_state_and_mean_aggregation_take_the_ranges_too():
    ei, n = union([lambda o: (_dense_graph(128, o), 128)] * 16)
    x, ea = inputs(n, ei.shape[1], 8)
    for over, opts in (({}, {"edge_state_dtype": "bf16"}), ({"node_agg_fn": "mean"}, {})):
        params, arch, sd = _default_model(1.0 / 127 if not over else 1.0, **over)
        a, b, state, flags = run_both(params, arch, sd, x, ei, ea, **opts)
        assert state == 0 and flags == 0
        for s, t in zip(a, b):
            assert torch.equal(s, t)


def test_one_step_forwards_never_ask():
    params, arch, sd = _default_model(1.0 / 63, num_enc_steps=1, num_class_steps=1)
    ei = _dense_graph(64)
    x, ea = inputs(64, ei.shape[1], 3)
    a, b, state, _ = run_both(params, arch, sd, x, ei, ea)
    assert state == 0
    for s, t in zip(a, b):
        assert torch.equal(s, t)

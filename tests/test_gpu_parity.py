"""Parity tests proper: the HIP path (through the C ABI, via gnn_cca_amd.MOTMPNet) against the golden vectors
produced by the reference and against the CPU oracle.  GPU only (`-m gpu`)."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_cases
from oracle.mpn_oracle import NumpyOracle, load_case

pytestmark = pytest.mark.gpu

TOL = 1e-4          # BASELINE.json north_star: logits within 1e-4 of the reference CPU path (fp32)
TOL_TIGHT = 5e-6    # what fp32 kernels actually achieve on the conditioned golden weights


class Data:
    def __init__(self, x, edge_index, edge_attr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr


def build(params, arch, sd):
    from gnn_cca_amd import MOTMPNet
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.cuda().eval()


def supported(m):
    import ctypes as C
    from gnn_cca_amd import _native as nat
    return nat.lib().gnncca_supported(C.byref(m.native_dims())) == 0


def to_data(a):
    return Data(torch.from_numpy(a["x"]).cuda(), torch.from_numpy(a["edge_index"]).cuda(),
                torch.from_numpy(a["edge_attr"]).cuda())


@pytest.mark.parametrize("name", golden_cases())
def test_hip_matches_reference_golden(name):
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = build(params, arch, sd)
    assert supported(m)
    trace = {}
    with torch.no_grad():
        out = m(to_data(a), trace=trace)["classified_edges"]
    torch.cuda.synchronize()
    assert len(out) == int(a["n_logits"])
    for i, o in enumerate(out):
        assert tuple(o.shape) == a[f"logits_{i}"].shape
        err = np.abs(o.cpu().numpy() - a[f"logits_{i}"]).max()
        assert err <= TOL_TIGHT, (name, i, err)
    # intermediate latents against the reference's own (debug taps)
    assert np.abs(trace["h_enc"].cpu().numpy() - a["h_enc"]).max() <= TOL_TIGHT
    assert np.abs(trace["e_enc"].cpu().numpy() - a["e_enc"]).max() <= TOL_TIGHT
    L = int(params["num_enc_steps"])
    for s in range(1, L + 1):
        assert np.abs(trace["e_steps"][s - 1].cpu().numpy() - a[f"e_step_{s}"]).max() <= TOL_TIGHT, (name, s)
        assert np.abs(trace["h_steps"][s - 1].cpu().numpy() - a[f"h_step_{s}"]).max() <= TOL_TIGHT, (name, s)
    assert m.graph_flags() in (0, 1)


@pytest.mark.parametrize("name", ["dense64", "ragged_sum", "union3"])
def test_untraced_equals_traced(name):
    """The production call (no trace; last step skips the dead node update) gives the same logits bit for bit."""
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = build(params, arch, sd)
    d = to_data(a)
    with torch.no_grad():
        o1 = [t.clone() for t in m(d)["classified_edges"]]
        o2 = m(d, trace={})["classified_edges"]
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)


def _dense_graph(n, offset=0):
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    k = i != j
    return np.stack([i[k] + offset, j[k] + offset]).astype(np.int64)


def _default_model(node_mlp_scale, seed=0, **over):
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    params = copy.deepcopy(params)
    params.update(over)
    sd = dict(sd)
    base = np.float32(63.0)  # dense64 stores node-MLP weights scaled by 1/63
    for k in list(sd):
        if k.startswith("MPNet.node_model"):
            sd[k] = (sd[k] * base * np.float32(node_mlp_scale)).astype(np.float32)
    return params, arch, sd


@pytest.mark.parametrize("n", [2, 33, 64, 65, 128, 256])
def test_dense_graphs_vs_oracle(n):
    """Full-size named configs (BASELINE.json configs 2/3) against the CPU oracle on the same seeded inputs."""
    params, arch, sd = _default_model(1.0 / max(n - 1, 1))
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = _dense_graph(n)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = build(params, arch, sd)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))
    for o, r in zip(out["classified_edges"], ref):
        err = np.abs(o.cpu().numpy() - r).max()
        assert err <= TOL_TIGHT * 4, (n, err)
        assert err <= TOL


def test_batch_of_graphs_equals_per_graph():
    """Disjoint union (Batch.from_data_list, inference.py:279) == each graph alone, bit for bit."""
    params, arch, sd = _default_model(1.0 / 40)
    m = build(params, arch, sd)
    rng = np.random.default_rng(3)
    sizes = [17, 64, 5, 90, 33]
    xs, eis, eas, off = [], [], [], 0
    for n in sizes:
        xs.append(rng.standard_normal((n, 2048)).astype(np.float32) / 45.0)
        eis.append(_dense_graph(n, off))
        eas.append(rng.random((eis[-1].shape[1], 4)).astype(np.float32))
        off += n
    with torch.no_grad():
        big = m(Data(torch.from_numpy(np.concatenate(xs)).cuda(), torch.from_numpy(np.concatenate(eis, 1)).cuda(),
                     torch.from_numpy(np.concatenate(eas)).cuda()))["classified_edges"]
        big = [b.clone() for b in big]
        e0, off = 0, 0
        for n, x, ei, ea in zip(sizes, xs, eis, eas):
            one = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei - off).cuda(),
                         torch.from_numpy(ea).cuda()))["classified_edges"]
            for b, o in zip(big, one):
                assert torch.equal(b[e0:e0 + ei.shape[1]], o), n
            e0 += ei.shape[1]
            off += n


def test_shuffled_edges_equivariant():
    """Unsorted `row` goes through the stable device sort; logits follow the edges (permutation equivariance)."""
    params, arch, sd = _default_model(1.0 / 47)
    m = build(params, arch, sd)
    rng = np.random.default_rng(9)
    n = 48
    x = rng.standard_normal((n, 2048)).astype(np.float32) / 45.0
    ei = _dense_graph(n)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    perm = rng.permutation(ei.shape[1])
    with torch.no_grad():
        a = [t.clone() for t in m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(),
                                        torch.from_numpy(ea).cuda()))["classified_edges"]]
        assert m.graph_flags() == 0
        b = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei[:, perm].copy()).cuda(),
                   torch.from_numpy(ea[perm].copy()).cuda()))["classified_edges"]
        assert m.graph_flags() == 1
    for s, t in zip(a, b):
        # the stable sort restores exactly the sorted graph's per-segment order only up to the shuffle within a
        # segment, so sums may differ in the last bits
        assert np.abs(s.cpu().numpy()[perm] - t.cpu().numpy()).max() <= TOL_TIGHT


def test_bad_index_poisons_outputs():
    params, arch, sd = _default_model(1.0 / 7)
    m = build(params, arch, sd)
    ei = _dense_graph(8)
    ei[1, 5] = 99  # out of range
    x = torch.randn(8, 2048).cuda()
    with torch.no_grad():
        out = m(Data(x, torch.from_numpy(ei).cuda(), torch.rand(ei.shape[1], 4).cuda()))["classified_edges"]
    assert m.graph_flags() & 2
    assert all(torch.isnan(o).all() for o in out)


def test_cpu_tensors_raise():
    params, arch, sd = _default_model(1.0)
    m = build(params, arch, sd)
    with pytest.raises(RuntimeError):
        m(Data(torch.randn(4, 2048), torch.zeros(2, 3, dtype=torch.long), torch.rand(3, 4)))


def test_empty_graph():
    params, arch, sd = _default_model(1.0)
    m = build(params, arch, sd)
    with torch.no_grad():
        out = m(Data(torch.randn(4, 2048).cuda(), torch.zeros(2, 0, dtype=torch.long).cuda(),
                     torch.rand(0, 4).cuda()))["classified_edges"]
    assert len(out) == 3 and all(tuple(o.shape) == (0, 1) for o in out)


def test_big_batch_split_bf16_encoder_vs_oracle():
    """64 x dense128 (N = 8192 nodes, E = 1.04 M edges: the per-GPU share of BASELINE config 4).  At this size the
    first encoder layer runs on the split-bf16 MFMA GEMM; its fp32-level accuracy is checked on the encoder output
    itself and on the logits."""
    params, arch, sd = _default_model(1.0 / 127)
    rng = np.random.default_rng(11)
    g, n = 64, 128
    x = rng.standard_normal((g * n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = np.concatenate([_dense_graph(n, k * n) for k in range(g)], axis=1)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    assert err_gpu <= max(4 * err_ref, 2e-7), (err_gpu, err_ref)  # as accurate as an fp32 GEMM, judged against fp64
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


@pytest.mark.parametrize("n_nodes", [4096 + 3, 6144, 6144 + 9, 8192 + 5, 16384, 16384 + 77, 40000, 51200 + 33])
def test_lds_staged_split_gemm_vs_fp64_oracle(n_nodes):
    """N >= 6144 nodes: the first encoder layer runs on the 256-row, both-operands-through-LDS split-bf16 GEMM, software-pipelined by
    k-half (ragged N exercises the clamped loads / masked stores; 6144 ... 40 000 split K and finish in the MFMA tail kernel over 8 or 4
    slabs, 51 233 runs un-split -- 201 row blocks, one round of workgroups -- with the rest of the encoder and the step-1
    projections in the GEMM's fused epilogue; 4 099 takes the 128-row GEMM and the wave-per-node tail).  A sparse ring
    graph keeps the oracle cheap; the encoder output is judged against an fp64 evaluation, the logits against the fp32
    oracle."""
    params, arch, sd = _default_model(1.0)
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    assert err_gpu <= max(4 * err_ref, 2e-7), (err_gpu, err_ref)
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


@pytest.mark.parametrize("n_nodes", [4096 + 5, 8192, 9000, 66000])
def test_fp16_split_gemm_short_k(n_nodes):
    """arch = 'bdnet_market' (node_in_dim 512): the fp16-split GEMMs with only 16 chunks of K -- the 32-row kernel's k-quarters are then four
    chunks long, SHORTER than its five-chunk x prefetch (clamped duplicate requests), and the 256-row kernel's split-K slices two chunks."""
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "bdnet512.npz"))
    rng = np.random.default_rng(n_nodes)
    k = params["encoder_feats_dict"]["nodes"][arch]["node_in_dim"]
    assert k == 512
    x = rng.standard_normal((n_nodes, k)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    assert err_gpu <= max(4 * err_ref, 2e-7), (err_gpu, err_ref)
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


@pytest.mark.parametrize("n_nodes", [301, 1229, 3000 + 7, 6144 + 9, 51200 + 33])
@pytest.mark.parametrize("what", ["x_beyond_fp16", "w_beyond_fp16", "x_tiny"])
def test_fp16_split_gemm_range_guard(n_nodes, what):
    """(301 / 1229 / 3007 nodes: round 6's 32-row slices kernel, csrc/enc_f16_slices.cuh, at 16 / 8 / 8 slices -- its arm is per WAVE, an
    exact-fp32 MFMA chain; the others: round 5's kernels.)
    The fp16-split GEMM (csrc/enc_f16.cuh, round 5) carries x and W as two fp16 pieces each.  fp16's exponent is narrow: a workgroup
    that meets a finite |x| >= 65520, or any workgroup when a weight is that large (flag words written by the packers), must recompute
    its tile on the bf16 six-product arm; magnitudes below fp16's normal range (6.1e-5) degrade gracefully (absolute error <= 2^-36 per
    element).  Encoder output against an fp64 evaluation -- RELATIVE to each row's magnitude where huge values are planted -- and logits
    against the fp32 oracle; split-K (6153 nodes) and fused un-split (51 233 nodes) launches."""
    params, arch, sd = _default_model(1.0)
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    hot = [5, 300, n_nodes - 2]
    if what == "x_beyond_fp16":
        x[hot[0], 7] = 7.0e4           # just beyond fp16 (65504)
        x[hot[1], 2040] = -3.0e9
        x[hot[2], 1000] = 65519.0      # still rounds to 65504: no arm needed for this one, and it must be right either way
    elif what == "w_beyond_fp16":
        sd["encoder.node_mlp.fc_layers.0.weight"][17, 33] = 1.0e5
    else:
        x[hot[0]] *= 1e-4              # a whole row far below fp16's normal range
        x[hot[1], ::2] = 0.0
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    got = trace["h_enc"].cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    scale = np.maximum(np.abs(h64).max(axis=1, keepdims=True), 1.0)          # per row: planted values make a row's outputs huge
    err_gpu = (np.abs(got - h64) / scale).max()
    err_ref = (np.abs(tr["h_enc"] - h64) / scale).max()
    assert err_gpu <= max(4 * err_ref, 2e-7), (err_gpu, err_ref)
    # logits: a planted magnitude of 1e4 ... 1e9 runs through sums whose terms cancel, so "close to the fp32 oracle" is the wrong yardstick
    # there (two fp32 evaluations differ by the rounding of the big intermediates): both are judged against an fp64 evaluation
    ref64 = NumpyOracle(params, arch, sd, np.float64).forward(x.astype(np.float64), ei, ea.astype(np.float64))
    for o, r, r64 in zip(out, ref, ref64):
        o = o.cpu().numpy().astype(np.float64)
        assert np.isfinite(o).all()
        s = np.maximum(np.abs(r64), 1.0)
        assert (np.abs(o - r64) / s).max() <= max(4 * (np.abs(r - r64) / s).max(), TOL_TIGHT * 2)


@pytest.mark.parametrize("n_nodes", [301, 1229, 6144 + 9, 51200 + 33])
def test_fp16_split_gemm_on_uniformly_tiny_inputs(n_nodes):
    """below fp16's normal range (6.1e-5) the two fp16 pieces of an operand carry 2^-36 ABSOLUTE precision, so an input that is
    tiny THROUGHOUT (un-normalised or rescaled features around 1e-6) would lose relative accuracy against an fp32 GEMM (1e-5 instead of
    6e-8).  Round 6: a wave / workgroup whose largest |x| is below 2^-10 without being zero takes the range arm (kF16Tiny, csrc/internal.h).
    x scaled by 1e-6 and both encoder biases zeroed, so that the encoder output is proportional to x: its error against an fp64 evaluation
    RELATIVE to the row's magnitude must stay within 4x the fp32 oracle's own -- at every kernel of the family (slices, 32-row, 256-row)."""
    params, arch, sd = _default_model(1.0)
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    sd["encoder.node_mlp.fc_layers.0.bias"][:] = 0.0
    sd["encoder.node_mlp.fc_layers.3.bias"][:] = 0.0
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    x *= np.float32(1e-6)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    tr = {}
    NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    trace = {}
    with torch.no_grad():
        m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()), trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    got = trace["h_enc"].cpu().numpy().astype(np.float64)
    scale = np.abs(h64).max(axis=1, keepdims=True)
    assert scale.min() > 0 and scale.max() < 1e-4          # the outputs ARE tiny: an absolute bound would say nothing
    err_gpu = (np.abs(got - h64) / scale).max()
    err_ref = (np.abs(tr["h_enc"] - h64) / scale).max()
    assert err_gpu <= max(4 * err_ref, 4e-7), (err_gpu, err_ref)


@pytest.mark.parametrize("n_nodes,products", [(4096 + 3, 6), (8192, 6), (8192 + 5, 6), (16384 + 77, 6), (30000, 6), (8192 + 5, 3)])
def test_unsplit_32_row_encoder_vs_fp64_oracle(n_nodes, products):
    """`model.encoder_unsplit = True`: mid-size batches take the un-split 32-row split-bf16 GEMM with the fused epilogue
    (csrc/enc_rows32.cuh; <= 256 workgroups: eight W stages per wave, more: four; ragged N exercises the clamped rows).  Encoder
    output against an fp64 evaluation, logits against the fp32 oracle, as for the other GEMMs."""
    params, arch, sd = _default_model(1.0)
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    m.encoder_unsplit = True
    m.encoder_products = products
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    if products == 6:
        assert err_gpu <= max(4 * err_ref, 2e-7), (err_gpu, err_ref)
    else:
        assert err_gpu <= 4e-5 * max(1.0, float(np.abs(h64).max())), err_gpu
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


@pytest.mark.parametrize("unsplit", [False, True])
def test_reattach_config_on_a_mid_size_batch(unsplit):
    """reattach_initial_nodes / _edges on 8200 nodes: the encoder's fused epilogues do not apply (the projections take cat(h0, h)),
    so the first layer runs split-K on the 256-row GEMM and finishes in the wave-per-node tail -- with and without
    `encoder_unsplit`, which such a configuration ignores.  Sparse ring graph, general step kernels, against the oracle."""
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "reattach_n1e1.npz"))
    n_nodes = 8200
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n_nodes, np.asarray(sd["encoder.node_mlp.fc_layers.0.weight"]).shape[1])).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], np.asarray(sd["encoder.edge_mlp.fc_layers.0.weight"]).shape[1])).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    m.encoder_unsplit = unsplit
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    assert err_gpu <= max(4 * err_ref, 2e-7 * max(1.0, float(np.abs(h64).max()))), (err_gpu, err_ref)
    for o, r in zip(out, ref):
        scale = max(1.0, float(np.abs(r).max()))
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2 * scale


def test_unsplit_encoder_launch_repairs_unsorted_plan():
    """The 32-row un-split launch folds the plan's findings in its extra workgroup, as the 256-row fused launch does: a SHUFFLED
    sparse edge list over 9000 nodes comes back in the caller's edge order."""
    params, arch, sd = _default_model(1.0)
    n_nodes = 9000
    rng = np.random.default_rng(98)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    perm = rng.permutation(src.shape[0])
    ei = np.ascontiguousarray(np.stack([src, dst])[:, perm]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = build(params, arch, sd)
    m.encoder_unsplit = True
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    assert m.graph_flags() & 1
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


@pytest.mark.parametrize("name", ["dense64", "terrace32", "union3", "ragged_mean", "dense24_shuffled", "steps_L8", "bdnet512"])
def test_bf16_edge_state_option(name):
    """GNNCCA_OPT_EDGE_STATE_BF16: edge latents stored as bf16 between steps, arithmetic fp32.  The logits must stay
    within north_star's 1e-4 of the reference; the measured deviation is reported in DESIGN.md."""
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = build(params, arch, sd)
    m.edge_state_dtype = "bf16"
    with torch.no_grad():
        out = m(to_data(a))["classified_edges"]
    worst = 0.0
    for i, o in enumerate(out):
        worst = max(worst, float(np.abs(o.cpu().numpy() - a[f"logits_{i}"]).max()))
    print(f"bf16 edge state, {name}: max |logit - reference| = {worst:.3e}")
    assert worst <= TOL
    m.edge_state_dtype = "fp32"
    with torch.no_grad():
        out32 = m(to_data(a))["classified_edges"]
    for i, o in enumerate(out32):
        assert np.abs(o.cpu().numpy() - a[f"logits_{i}"]).max() <= TOL_TIGHT


@pytest.mark.parametrize("name", [n for n in golden_cases() if not n.startswith("generic_")])
def test_device_pack_equals_host_pack(name):
    """gnncca_pack_weights_device (what a training step / load_state_dict uses) builds the host packer's blob byte for
    byte: BatchNorm fold in double, split-weight layouts, bf16 planes."""
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = build(params, arch, sd)
    dev_blob = m._pack_weights_device(torch.device("cuda", torch.cuda.current_device()))
    assert dev_blob is not None, "tuned family must pack on the GPU"
    torch.cuda.synchronize()
    assert torch.equal(dev_blob.cpu(), m.pack_weights_host()), name


def test_device_repack_follows_parameter_updates():
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    m = build(params, arch, sd)
    data = to_data(a)
    with torch.no_grad():
        before = m(data)["classified_edges"][-1].clone()
        for p in m.parameters():
            p.mul_(1.01)            # in-place update, like an optimizer step: bumps the tensors' version counters
        after = m(data)["classified_edges"][-1]
        m2 = build(params, arch, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()})
        assert torch.equal(m._packed[1].cpu(), m2.pack_weights_host())
        assert not torch.equal(before, after)
        assert torch.allclose(after, m2(data)["classified_edges"][-1], atol=0, rtol=0)


def test_concurrent_streams_one_module():
    """One module, three streams, three different graphs in flight at once (a workspace per stream, shared packed
    weights): every stream's logits equal the ones the same graph produces alone."""
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    m = build(params, arch, sd)
    rng = np.random.default_rng(3)
    datas = []
    for s in range(3):
        x = a["x"] + (0.01 * s) * rng.standard_normal(a["x"].shape).astype(np.float32)
        ea = rng.random(a["edge_attr"].shape).astype(np.float32)
        datas.append(Data(torch.from_numpy(x).cuda(), torch.from_numpy(a["edge_index"]).cuda(), torch.from_numpy(ea).cuda()))
    with torch.no_grad():
        alone = [[t.clone() for t in m(d)["classified_edges"]] for d in datas]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in datas]
        outs = [None] * 3
        for rep in range(20):  # interleaved issue: the three forwards overlap on the GPU
            for s, (d, st) in enumerate(zip(datas, streams)):
                with torch.cuda.stream(st):
                    outs[s] = m(d)["classified_edges"]
        torch.cuda.synchronize()
    assert len(m._workspaces) >= 3
    for s in range(3):
        for o, r in zip(outs[s], alone[s]):
            assert torch.equal(o, r), s


# ---- padded edge-state layout of big, nearly regular batches (StepParams::ell_S, csrc/step_fast.cuh) -------------------
def _union(sizes, rng, extra=None):
    """Disjoint union of dense graphs of the given sizes (+ optional extra (x-rows, edge list) appended at the end)."""
    xs, eis, off = [], [], 0
    for n in sizes:
        eis.append(_dense_graph(n, off))
        off += n
    n_extra = 0
    if extra is not None:
        n_extra, e_extra = extra
        eis.append(e_extra + off)
        off += n_extra
    x = rng.standard_normal((off, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = np.concatenate(eis, axis=1).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    return x, ei, ea


@pytest.mark.parametrize("agg,edge_state", [("sum", "fp32"), ("mean", "fp32"), ("sum", "bf16")])
def test_padded_layout_ragged_batch_vs_oracle(agg, edge_state):
    """45 dense graphs of 120..129 nodes (E ~ 0.69 M >= 2^19, degrees 119..128, 97 % fill): the step kernels keep the edge
    state in the padded layout (128 slots per node).  Degrees differ from node to node, every segment ends in padding."""
    params, arch, sd = _default_model(1.0 / 124, node_agg_fn=agg)
    rng = np.random.default_rng(5)
    sizes = [120 + (7 * g) % 10 for g in range(45)]
    x, ei, ea = _union(sizes, rng)
    assert ei.shape[1] >= 1 << 19
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = build(params, arch, sd)
    m.edge_state_dtype = edge_state
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    assert m.graph_flags() == 0
    tol = TOL if edge_state == "bf16" else TOL_TIGHT * 2
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= tol


def test_padded_layout_overflow_falls_back_to_compact_order():
    """60 dense128 graphs + one hub with 400 out-edges to 400 leaves (which have none): E/N still says 128 slots per node,
    the hub does not fit.  The plan raises GNNCCA_GRAPH_IRREGULAR (bit 2 of the flag word, informational) and every step
    uses the compact order -- same logits as the oracle; zero-degree nodes included."""
    params, arch, sd = _default_model(1.0 / 127)
    rng = np.random.default_rng(6)
    hub = np.stack([np.zeros(400, dtype=np.int64), np.arange(1, 401, dtype=np.int64)])
    x, ei, ea = _union([128] * 60, rng, extra=(401, hub))
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = build(params, arch, sd)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    assert m.graph_flags() == 4
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


def test_padded_layout_unsorted_rows_fall_back():
    """The same kind of batch with its edge list shuffled: UNSORTED -> the stable device sort + compact order."""
    params, arch, sd = _default_model(1.0 / 127)
    rng = np.random.default_rng(7)
    x, ei, ea = _union([128] * 36, rng)
    perm = rng.permutation(ei.shape[1])
    ei, ea = np.ascontiguousarray(ei[:, perm]), np.ascontiguousarray(ea[perm])
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = build(params, arch, sd)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    assert m.graph_flags() & 1
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 4


@pytest.mark.parametrize("case", ["sorted", "shuffled", "odd_E", "bad_index", "ragged"])
def test_plan_on_mid_size_batches(case):
    """Batches of 6144 ... 38 399 nodes with >= 2^19 edges: the first encoder layer runs split-K on the 256-row GEMM, the graph plan
    takes the pair form in a launch of its own and the tail launch folds its findings.  52 dense graphs (N = 6656, E = 845 312):
    sorted; shuffled (UNSORTED -> stable device sort); one edge dropped (odd E: the narrow form); an index out of range (poisoned
    logits); ragged sizes."""
    params, arch, sd = _default_model(1.0 / 127)
    rng = np.random.default_rng(21)
    sizes = [128] * 52 if case != "ragged" else [100 + (13 * g) % 60 for g in range(56)]
    x, ei, ea = _union(sizes, rng)
    assert x.shape[0] >= 6144 and ei.shape[1] >= 1 << 19
    if case == "shuffled":
        perm = rng.permutation(ei.shape[1])
        ei, ea = np.ascontiguousarray(ei[:, perm]), np.ascontiguousarray(ea[perm])
    if case == "odd_E":
        keep = np.ones(ei.shape[1], bool)
        keep[12345] = False
        ei, ea = np.ascontiguousarray(ei[:, keep]), np.ascontiguousarray(ea[keep])
        assert ei.shape[1] % 2 == 1
    m = build(params, arch, sd)
    if case == "bad_index":
        ei = ei.copy()
        ei[1, 700001] = x.shape[0] + 5
        with torch.no_grad():
            out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
        assert m.graph_flags() & 2
        assert all(torch.isnan(o).all() for o in out)
        return
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    assert (m.graph_flags() & 1) == (1 if case == "shuffled" else 0)
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * (4 if case == "shuffled" else 2)


def test_fused_encoder_launch_repairs_unsorted_plan():
    """N = 51 233 with a SHUFFLED sparse edge list: the fused GEMM launch's extra workgroup folds the plan's findings and
    runs the stable sort itself (there is no tail launch in this regime); logits in the caller's edge order."""
    params, arch, sd = _default_model(1.0)
    n_nodes = 51200 + 33
    rng = np.random.default_rng(99)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    perm = rng.permutation(src.shape[0])
    ei = np.ascontiguousarray(np.stack([src, dst])[:, perm]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = build(params, arch, sd)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    assert m.graph_flags() & 1
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2


@pytest.mark.parametrize("n_nodes", [8192 + 5, 16384 + 77, 51200 + 33])
def test_encoder_three_product_option(n_nodes):
    """GNNCCA_OPT_ENC_SPLIT3 (model.encoder_products = 3): the first encoder layer keeps the three leading split-bf16 products.
    The logits must stay within north_star's 1e-4 of the fp32 oracle (measured: 1.5e-7); the encoder output is allowed the
    ~2^-17 relative error of the dropped terms (measured 5e-6 on O(1) activations), not more."""
    params, arch, sd = _default_model(1.0)
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, 2048)).astype(np.float32)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    m = build(params, arch, sd)
    m.encoder_products = 3
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    err_h = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    assert err_h <= 4e-5 * max(1.0, float(np.abs(h64).max())), err_h
    for o, r in zip(out, ref):
        err = np.abs(o.cpu().numpy() - r).max()
        assert err <= TOL_TIGHT, err          # far inside the 1e-4 of north_star


@pytest.mark.parametrize("n,L", [(256, 4), (1024, 8)])
def test_bf16_edge_state_at_named_sizes(n, L):
    """BASELINE configs 3 and 5 name bf16: dense 256 / L = 4 and dense 1024 / L = 8 with the edge latents stored as bf16
    between steps (arithmetic fp32), against the fp32 CPU oracle at FULL size.  Tolerance: north_star's 1e-4 absolute on
    logits of O(1); the measured absolute and relative deviations are printed (pytest -s) and quoted in DESIGN.md."""
    params, arch, sd = _default_model(1.0 / (n - 1), num_enc_steps=L)
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = _dense_graph(n)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    worst = {}
    for es in ("fp32", "bf16"):
        m = build(params, arch, sd)
        m.edge_state_dtype = es
        with torch.no_grad():
            out = m(d)["classified_edges"]
        err = max(float(np.abs(o.cpu().numpy() - r).max()) for o, r in zip(out, ref))
        scale = max(float(np.abs(r).max()) for r in ref)
        worst[es] = (err, err / scale)
    print(f"dense{n} L={L}: max |dlogit| fp32 state {worst['fp32'][0]:.2e} (rel {worst['fp32'][1]:.2e}), "
          f"bf16 state {worst['bf16'][0]:.2e} (rel {worst['bf16'][1]:.2e})")
    assert worst["fp32"][0] <= TOL_TIGHT * 4
    assert worst["bf16"][0] <= TOL


@pytest.mark.parametrize("name", ["terrace32", "reattach_n1e1", "steps_L0", "generic_dims"])
def test_forward_hooks_on_containers_see_reference_tensors(name):
    """Forward hooks on model.encoder / model.MPNet / model.classifier -- how per-step latents are tapped from the reference
    (tests/golden/make_golden.py does exactly that) -- fire with the reference's inputs and outputs: the module replays the
    call sequence of models/mpn.py:266-297 over the traced native forward.  Hooked and un-hooked logits are identical."""
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = build(params, arch, sd)
    d = to_data(a)
    with torch.no_grad():
        plain = [t.clone() for t in m(d)["classified_edges"]]
    seen = {"enc": [], "mp_in": [], "mp_out": [], "cls": []}
    hooks = [m.encoder.register_forward_hook(lambda mod, i, o: seen["enc"].append(o)),
             m.MPNet.register_forward_pre_hook(lambda mod, i: seen["mp_in"].append(i)),
             m.MPNet.register_forward_hook(lambda mod, i, o: seen["mp_out"].append(o)),
             m.classifier.register_forward_hook(lambda mod, i, o: seen["cls"].append(o))]
    with torch.no_grad():
        out = m(d)["classified_edges"]
    for h in hooks:
        h.remove()
    L = int(params["num_enc_steps"])
    assert len(seen["enc"]) == 1 and len(seen["mp_out"]) == L and len(seen["cls"]) == len(out) == int(a["n_logits"])
    e_enc, h_enc = seen["enc"][0]                      # (edge_out, node_out): edge first (models/mpn.py:142)
    assert np.abs(e_enc.cpu().numpy() - a["e_enc"]).max() <= TOL_TIGHT
    assert np.abs(h_enc.cpu().numpy() - a["h_enc"]).max() <= TOL_TIGHT
    for s in range(1, L + 1):
        x_in, ei_in, e_in = seen["mp_in"][s - 1]
        nf = 2 if params["reattach_initial_nodes"] else 1
        ef = 2 if params["reattach_initial_edges"] else 1
        assert x_in.shape[1] == nf * h_enc.shape[1] and e_in.shape[1] == ef * e_enc.shape[1] and ei_in is d.edge_index
        h_s, e_s = seen["mp_out"][s - 1]               # (x, edge_attr) (models/mpn.py:54)
        assert np.abs(h_s.cpu().numpy() - a[f"h_step_{s}"]).max() <= TOL_TIGHT
        assert np.abs(e_s.cpu().numpy() - a[f"e_step_{s}"]).max() <= TOL_TIGHT
    for (dec, none), o, p in zip(seen["cls"], out, plain):
        assert none is None and dec is o
        assert np.abs(o.cpu().numpy() - p.cpu().numpy()).max() <= TOL_TIGHT
    # a container on its own, outside MOTMPNet.forward: evaluates itself (test_submodules_called_on_their_own)
    if L > 0:
        h1, e1 = m.MPNet(seen["mp_in"][0][0], d.edge_index, seen["mp_in"][0][2])
        assert np.abs(h1.cpu().numpy() - a["h_step_1"]).max() <= 2e-5 and np.abs(e1.cpu().numpy() - a["e_step_1"]).max() <= 2e-5


def test_wrong_input_dtypes_raise_like_the_reference():
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, "terrace32.npz"))
    m = build(params, arch, sd)
    d = to_data(a)
    for bad in (Data(d.x.double(), d.edge_index, d.edge_attr), Data(d.x, d.edge_index, d.edge_attr.half()),
                Data(d.x, d.edge_index.int(), d.edge_attr)):
        with pytest.raises(RuntimeError):
            m(bad)
        m.train()
        with pytest.raises(RuntimeError):
            m(bad)
        m.eval()
    with torch.no_grad():                              # a non-contiguous view of the right dtype is only a layout
        xt = d.x.t().contiguous().t()
        assert not xt.is_contiguous()
        out = m(Data(xt, d.edge_index, d.edge_attr))["classified_edges"]
    assert np.abs(out[-1].cpu().numpy() - a[f"logits_{int(a['n_logits']) - 1}"]).max() <= TOL_TIGHT


@pytest.mark.parametrize("name", ["terrace32", "ragged_max", "ragged_mean", "reattach_n1e1", "generic_dims", "generic_reattach_max",
                                  "dense24_shuffled", "steps_L0"])
def test_submodules_called_on_their_own(name):
    """The reference's sub-modules are callable by themselves (models/mpn.py:128-142 encoder / classifier, :32-54 MetaLayer,
    :59-69 EdgeModel, :71-101 NodeModel, models/mlp.py:26-28 MLP).  Here they evaluate through gnncca_mlp_eval /
    gnncca_gather_cat / gnncca_aggregate (eval mode); chained by hand as models/mpn.py:266-297 chains them they reproduce the
    reference's golden latents and logits."""
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = build(params, arch, sd)
    d = to_data(a)
    tol = 2e-5
    with torch.no_grad():
        e_enc, h_enc = m.encoder(d.edge_attr, d.x)              # edge first (mpn.py:142)
        assert np.abs(h_enc.cpu().numpy() - a["h_enc"]).max() <= tol and np.abs(e_enc.cpu().numpy() - a["e_enc"]).max() <= tol
        assert m.encoder(None, d.x)[0] is None                   # None passes through (mpn.py:133-140)
        h, e = h_enc, e_enc
        L, first = int(params["num_enc_steps"]), int(params["num_enc_steps"]) - int(params["num_class_steps"]) + 1
        logits = []
        for step in range(1, L + 1):
            if params["reattach_initial_edges"]:
                e = torch.cat((e_enc, e), dim=1)
            if params["reattach_initial_nodes"]:
                h = torch.cat((h_enc, h), dim=1)
            if step == 1:   # the two halves of a MetaLayer by hand (mpn.py:48,52)
                row, col = d.edge_index
                e_new = m.MPNet.edge_model(h[row], h[col], e)
                h_new = m.MPNet.node_model(h, d.edge_index, e_new)
                h2, e2 = m.MPNet(h, d.edge_index, e)
                assert torch.equal(h2, h_new) and torch.equal(e2, e_new)
                h, e = h_new, e_new
            else:
                h, e = m.MPNet(h, d.edge_index, e)
            assert np.abs(h.cpu().numpy() - a[f"h_step_{step}"]).max() <= tol, (name, step)
            assert np.abs(e.cpu().numpy() - a[f"e_step_{step}"]).max() <= tol, (name, step)
            if step >= first:
                dec, none = m.classifier(e)
                assert none is None
                logits.append(dec)
        if L == 0:
            logits.append(m.classifier(e)[0])
        assert len(logits) == int(a["n_logits"])
        for i, t in enumerate(logits):
            assert t.shape == a[f"logits_{i}"].shape and np.abs(t.cpu().numpy() - a[f"logits_{i}"]).max() <= tol, (name, i)
        # a bare MLP
        y = m.encoder.node_mlp(d.x)
        assert torch.equal(y, h_enc)
    m.train()
    with pytest.raises(RuntimeError):                            # stand-alone calls are eval-only
        m.encoder(d.edge_attr, d.x)
    m.eval()
    with pytest.raises(RuntimeError):                            # no CPU fallback
        m.encoder.node_mlp(d.x.cpu())
    with pytest.raises(RuntimeError):                            # wrong dtype raises like the reference's Linear
        m.encoder.node_mlp(d.x.double())


@pytest.mark.parametrize("node_in,n_nodes", [(512, 8192 + 7), (512, 20000), (96, 7000), (64, 6144)])
def test_pipelined_gemm_short_k(node_in, n_nodes):
    """Narrow node features on a batch (arch 'bdnet_market' has node_in 512): the pipelined 256-row GEMM then runs very few 32-deep
    chunks per workgroup (K / split-K / 32 = 8 ... 1) -- its prologue, its single-chunk tail and the clamped duplicate loads."""
    from gnn_cca_amd import MOTMPNet
    params, arch, _ = _default_model(1.0)
    params = copy.deepcopy(params)
    params["encoder_feats_dict"]["nodes"][arch]["node_in_dim"] = node_in
    torch.manual_seed(node_in + n_nodes)
    ref_m = MOTMPNet(copy.deepcopy(params), None, arch)
    sd = {k: v.detach().clone().numpy() for k, v in ref_m.state_dict().items()}
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, node_in)).astype(np.float32)
    src = np.repeat(np.arange(n_nodes), 2)
    dst = (src + np.tile([1, 5], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    assert err_gpu <= max(4 * err_ref, 2e-7 * max(1.0, float(np.abs(h64).max()))), (err_gpu, err_ref)
    for o, r in zip(out, ref):
        scale = max(1.0, float(np.abs(r).max()))
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2 * scale


@pytest.mark.parametrize("node_in,n_nodes", [(2048, 384), (2048, 400), (2048, 700), (2048, 1000), (512, 384), (512, 999), (96, 500), (64, 640),
                                              (2048, 2559), (2048, 2560), (2048, 3001), (512, 3500)])
def test_split_gemm_on_graphs_of_a_few_hundred_nodes(node_in, n_nodes):
    """From 384 nodes the first encoder layer runs on the 128-row split-bf16 GEMM with the plan riding in its launch (round 3; below:
    the f32 MFMA GEMM).  Few row blocks, deep split-K (up to 32 slices of 64 k), short K (one to three 32-deep chunks per workgroup),
    ragged N; the register-resident tail sums the slabs (below 2560 nodes; from there the MFMA tail, 32 nodes per workgroup).  Encoder
    output against fp64, logits against the fp32 oracle."""
    from gnn_cca_amd import MOTMPNet
    params, arch, _ = _default_model(1.0)
    params = copy.deepcopy(params)
    params["encoder_feats_dict"]["nodes"][arch]["node_in_dim"] = node_in
    torch.manual_seed(node_in + n_nodes)
    ref_m = MOTMPNet(copy.deepcopy(params), None, arch)
    sd = {k: v.detach().clone().numpy() for k, v in ref_m.state_dict().items()}
    rng = np.random.default_rng(n_nodes)
    x = rng.standard_normal((n_nodes, node_in)).astype(np.float32)
    src = np.repeat(np.arange(n_nodes), 3)
    dst = (src + np.tile([1, 5, 17], n_nodes)) % n_nodes
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    err_gpu = np.abs(trace["h_enc"].cpu().numpy() - h64).max()
    err_ref = np.abs(tr["h_enc"] - h64).max()
    assert err_gpu <= max(4 * err_ref, 2e-7 * max(1.0, float(np.abs(h64).max()))), (err_gpu, err_ref)
    for o, r in zip(out, ref):
        scale = max(1.0, float(np.abs(r).max()))
        assert np.abs(o.cpu().numpy() - r).max() <= TOL_TIGHT * 2 * scale


def test_non_finite_inputs_stay_inside_their_graph():
    """Non-finite node features are outside the contract of the fast kernels (their ReLU is an integer max on the float's bits:
    a positive NaN propagates, a negative one becomes 0; torch propagates both) -- what IS pinned: they do not leak.  A NaN / Inf
    graph in a batch affects its own logits only; every other graph's logits are bit for bit what they are next to a clean
    neighbour (lanes beyond a segment read zeros through the range-checked buffer descriptors, never a neighbour's values)."""
    params, arch, sd = _default_model(1.0 / 63)
    rng = np.random.default_rng(21)
    n, g = 64, 3
    x = rng.standard_normal((g * n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = np.concatenate([_dense_graph(n, k * n) for k in range(g)], axis=1)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    m = build(params, arch, sd)
    e_per = n * (n - 1)

    def run(xx, aa):
        with torch.no_grad():
            return [t.clone() for t in m(Data(torch.from_numpy(xx).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(aa).cuda()))["classified_edges"]]

    clean = run(x, ea)
    bad_x, bad_a = x.copy(), ea.copy()
    bad_x[n + 3, 17] = np.nan            # graph 1: a NaN feature, an infinite one, a NaN edge attribute
    bad_x[n + 9, 5] = np.inf
    bad_a[e_per + 11, 2] = np.nan
    dirty = run(bad_x, bad_a)
    for c, d in zip(clean, dirty):
        assert torch.equal(c[:e_per], d[:e_per]) and torch.equal(c[2 * e_per:], d[2 * e_per:])   # graphs 0 and 2: untouched
    # (graph 1 itself: unspecified -- the kernels' fmaxf / integer-max ReLUs swallow some NaNs that torch would propagate)


@pytest.mark.parametrize("n,g", [(100, 7), (450, 2)])
def test_non_finite_node_zero_stays_inside_graph_zero_on_the_buffer_addressed_kernel(n, g):
    """The same property on mpn_step_pipe_kernel (batches beyond 512 nodes; 2 x 450 nodes: its LDS gather-table variant), with the
    poison in NODE 0 of graph 0.  A lane beyond its segment reads target id 0 from its out-of-range index load; if it then gathered
    node 0's projection row, an Inf there would reach -- through the additive -3e38 tail mask of the split-bf16 message (inf - inf) --
    the sums of every node with a partial chunk, in every graph.  Degrees 99 / 449 leave a partial chunk in every segment."""
    params, arch, sd = _default_model(1.0 / (n - 1))
    rng = np.random.default_rng(23)
    x = rng.standard_normal((g * n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = np.concatenate([_dense_graph(n, k * n) for k in range(g)], axis=1)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    m = build(params, arch, sd)
    e_per = n * (n - 1)

    def run(xx):
        with torch.no_grad():
            return [t.clone() for t in m(Data(torch.from_numpy(xx).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]]

    clean = run(x)
    bad = x.copy()
    bad[0, 3] = np.inf                   # graph 0, node 0
    bad[0, 40] = np.nan
    dirty = run(bad)
    for c, d in zip(clean, dirty):
        assert torch.isfinite(c).all()
        assert torch.equal(c[e_per:], d[e_per:])          # graphs 1 ... g-1: bit for bit what they are next to a clean neighbour


@pytest.mark.parametrize("kind", ["ragged", "ragged_unsorted", "regular", "empty_tail"])
def test_ragged_batches_whose_edge_count_is_a_multiple_of_the_node_count(kind):
    """Batches beyond 1024 nodes (gather table in HBM, buffer-addressed step kernel) with E = 20 N but out-degrees 10 and 30 (and a
    shuffled edge list; and trailing nodes without edges; and the truly regular case): nothing may be inferred from E / N.  (A form
    of the step kernel that requested its first round on the assumption "every degree = E / N" and checked it afterwards was
    built against this test and measured: no gain, profiles/r03_logs/r3_ab_spec1.log; not kept.)  Against the traced forward
    (general kernel) bit for bit, and against the oracle."""
    params, arch, sd = _default_model(1.0 / 20)
    rng = np.random.default_rng(31)
    n = 2048
    if kind == "regular":
        deg = np.full(n, 20)
    elif kind == "empty_tail":
        deg = np.concatenate([np.full(n // 2, 40), np.zeros(n - n // 2, dtype=np.int64)])
    else:
        deg = np.where(np.arange(n) % 2 == 0, 10, 30)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, size=rows.size)
    assert rows.size % n == 0
    ei = np.stack([rows, cols]).astype(np.int64)
    if kind == "ragged_unsorted":
        ei = ei[:, rng.permutation(ei.shape[1])]
    x = rng.standard_normal((n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    with torch.no_grad():
        fast = [t.clone() for t in m(d)["classified_edges"]]
        traced = m(d, trace={})["classified_edges"]
    for a, b in zip(fast, traced):
        assert torch.equal(a, b), kind
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    for a, r in zip(fast, ref):
        assert np.abs(a.cpu().numpy() - r).max() <= TOL_TIGHT * 2, kind
    assert m.graph_flags() == (1 if kind == "ragged_unsorted" else 0)

"""Kernel variants that no default dispatch reaches any more keep their coverage here:
  * the generic family's op-by-op eval forward (rounds 1-3), now the fallback for widths beyond the fused step's LDS budget -- reached
    with a node latent of 160 -- and, through GNNCCA_GEN_UNFUSED, comparable with the fused form on the same configuration;
  * the step kernels' LDS gather table (off by default since round 4), reached through GNNCCA_PD_LDS_MIN in a child process (the
    library reads its diagnostic switches once per process)."""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, ROOT
from oracle.mpn_oracle import NumpyOracle, load_case
from test_gpu_parity import Data, _dense_graph, build

pytestmark = pytest.mark.gpu


def _wide_params(latent):
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    params = copy.deepcopy(params)
    params["encoder_feats_dict"]["nodes"][arch]["node_out_dim"] = latent
    params["node_model_feats_dict"]["fc_dims"] = [latent]
    return params, arch


def _random_model(params, arch, n, seed=0):
    from gnn_cca_amd import MOTMPNet
    torch.manual_seed(seed)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    with torch.no_grad():
        for p in m.MPNet.node_model.node_mlp.parameters():
            p.mul_(1.0 / max(n - 1, 1))
    sd = {k: v.detach().clone().numpy() for k, v in m.state_dict().items()}
    return m.cuda().eval(), sd


@pytest.mark.parametrize("latent", [160, 48])
def test_generic_family_wide_and_narrow_vs_oracle(latent):
    """node latent 160: beyond the fused step's per-thread LDS budget -> the op-by-op path; 48: the fused path.  Both against the
    oracle on a dense graph and a ragged union."""
    params, arch = _wide_params(latent)
    n = 40
    m, sd = _random_model(params, arch, n)
    rng = np.random.default_rng(3)
    ei = np.concatenate([_dense_graph(n), _dense_graph(17, n)[:, ::3]], axis=1)
    nn = n + 17
    x = rng.standard_normal((nn, 2048)).astype(np.float32) / 45.0
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= 2e-5 * max(1.0, float(np.abs(r).max()))



@pytest.mark.parametrize("edge_fc,node_fc", [([64, 48, 6], [64, 64]), ([128, 6], [128, 64])])
def test_generic_family_lds_budget_picks_a_tile_that_fits(edge_fc, node_fc):
    """node latent 64 with multi-layer MPN MLPs on a graph of <= 1024 nodes.  The widest tile (T = 256) of the fused generic
    step needs 2 * 64 * 257 * 4 B = 131.6 KB of activations plus ~36 KB of staged weights there -- more than a CU's 160 KB; a 128-wide
    layer needs 132 KB + weights at T = 128 too.  `gen_fused_ok` now sizes the WHOLE footprint and steps down to the 128-edge tile or to
    the op-by-op path instead of letting the launch fail (GNNCCA_ERR_HIP)."""
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    params = copy.deepcopy(params)
    params["encoder_feats_dict"]["nodes"][arch]["node_out_dim"] = 64
    params["edge_model_feats_dict"]["fc_dims"] = list(edge_fc)
    params["node_model_feats_dict"]["fc_dims"] = list(node_fc)
    n = 40
    m, sd = _random_model(params, arch, n)
    rng = np.random.default_rng(5)
    ei = np.concatenate([_dense_graph(n), _dense_graph(17, n)[:, ::3]], axis=1)
    nn = n + 17
    x = rng.standard_normal((nn, 2048)).astype(np.float32) / 45.0
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    for o, r in zip(out, ref):
        assert np.isfinite(o.cpu().numpy()).all()
        assert np.abs(o.cpu().numpy() - r).max() <= 2e-5 * max(1.0, float(np.abs(r).max()))


CHILD = r"""
import json, os, sys, copy
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
from oracle.mpn_oracle import load_case
from test_gpu_parity import Data, _default_model, _dense_graph, build
what = sys.argv[2]
res = {}
if what == "pd_lds":
    for n, g in ((900, 1), (300, 3), (64, 1)):
        params, arch, sd = _default_model(1.0 / (n - 1))
        rng = np.random.default_rng(n)
        ei = np.concatenate([_dense_graph(n, k * n) for k in range(g)], axis=1)
        x = rng.standard_normal((g * n, 2048)).astype(np.float32); x /= np.linalg.norm(x, axis=0, keepdims=True)
        ea = rng.random((ei.shape[1], 4)).astype(np.float32)
        m = build(params, arch, sd)
        with torch.no_grad():
            out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
        res[f"{g}x{n}"] = [o.cpu().numpy().tolist() for o in out][-1][:2000]
elif what == "npw":
    import hashlib
    # batches of 3000 ... 5000 nodes with <= 2 chunks of 64 edges per node: the shapes the two-nodes-per-wave step kernels accept
    for seed, (n, g, drop) in enumerate(((100, 36, 0), (120, 30, 7), (90, 50, 3))):
        params, arch, sd = _default_model(1.0 / (n - 1))
        rng = np.random.default_rng(100 + seed)
        parts = []
        for k in range(g):
            e = _dense_graph(n, k * n)
            if drop:                                        # ragged: every drop-th edge removed, one node left without out-edges
                keep = (np.arange(e.shape[1]) % drop != 0) & (e[0] != k * n + 5)
                e = e[:, keep]
            parts.append(e)
        ei = np.concatenate(parts, axis=1)
        x = rng.standard_normal((g * n, 2048)).astype(np.float32); x /= np.linalg.norm(x, axis=0, keepdims=True)
        ea = rng.random((ei.shape[1], 4)).astype(np.float32)
        m = build(params, arch, sd)
        with torch.no_grad():
            out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
        assert all(np.isfinite(o.cpu().numpy()).all() for o in out)
        res[f"{g}x{n}/{drop}"] = hashlib.sha256(b"".join(o.cpu().numpy().tobytes() for o in out)).hexdigest()
else:
    params, arch, sd, a = load_case(os.path.join(sys.argv[1], "tests", "golden", "generic_dims.npz"))
    m = build(params, arch, sd)
    with torch.no_grad():
        out = m(Data(*(torch.from_numpy(a[k]).cuda() for k in ("x", "edge_index", "edge_attr"))))["classified_edges"]
    res["generic_dims"] = [o.cpu().numpy().tolist() for o in out][-1]
    res["golden_err"] = max(float(np.abs(o.cpu().numpy() - a[f"logits_{i}"]).max()) for i, o in enumerate(out))
print("RESULT " + json.dumps(res))
"""


def _child(what, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, what], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_lds_gather_table_variant_gives_the_same_bits():
    """GNNCCA_PD_LDS_MIN=1 stages the P_dst table in LDS for every graph of <= 1024 nodes (rounds 1-2's default); the gathers return
    the same rows whichever memory they come from: bitwise-equal logits on the f32 kernel (64, 3 x 300 nodes) and the buffer-addressed
    one (900 nodes)."""
    base = _child("pd_lds", {})
    lds = _child("pd_lds", {"GNNCCA_DIAG": "1", "GNNCCA_PD_LDS_MIN": "1", "GNNCCA_PD_LDS_MAX": "1024"})
    assert sorted(base) == sorted(lds)
    for k in base:
        assert base[k] == lds[k], k


def test_one_and_two_nodes_per_wave_give_the_same_bits():
    """The message steps of batches below 16 384 nodes run one node per wave (since round 5 with the ids-first prologue), above that two
    nodes per wave (round 4's prologue order; GNNCCA_NPW = 2 forces that form wherever it is eligible).  Same arithmetic in the same order per
    node: the logits of three 3 000 ... 4 500-node batches -- regular and ragged, one node without out-edges -- hash the same either way."""
    one = _child("npw", {"GNNCCA_DIAG": "1", "GNNCCA_NPW": "1"})
    two = _child("npw", {"GNNCCA_DIAG": "1", "GNNCCA_NPW": "2"})
    assert sorted(one) == sorted(two) and len(one) == 3
    for k in one:
        assert one[k] == two[k], k


def test_generic_op_by_op_path_agrees_with_the_fused_step():
    fused = _child("generic", {})
    unfused = _child("generic", {"GNNCCA_DIAG": "1", "GNNCCA_GEN_UNFUSED": "1"})
    assert fused["golden_err"] <= 5e-6 and unfused["golden_err"] <= 5e-6
    assert np.abs(np.asarray(fused["generic_dims"]) - np.asarray(unfused["generic_dims"])).max() <= 5e-6


def test_union_beyond_the_buffer_addressed_step_kernel():
    """`csrc/mpn_forward.hip` switches the node message to the f32 form (`StepParams::msg_f32`) when `step_pipe_fits`
    fails -- a union whose edge state no longer sits below the 2^31-byte reach of the buffer-addressed step kernel (6 planes x e_stride x 4 B
    > 2^31: beyond ~ 89 M edge slots).  5 600 x dense128 = 91.0 M edges crosses it.  Inputs are generated on the device (13 GB in all; the
    host sees a sample only).  Checked: every logit finite; the first, a middle and the last graphs of the union within 2e-6 of the SAME
    graphs run as a small batch (which takes the split-bf16 message form and other encoder / step regimes) -- the bound
    tests/test_gpu_sharded.py states for a graph's logits across dispatch regimes -- and that small batch against the fp32 oracle."""
    import bench
    from test_gpu_parity import _default_model
    n, g = 128, 5600
    params, arch, sd = _default_model(1.0 / (n - 1))
    m = build(params, arch, sd)
    dev = torch.device("cuda", 0)
    ei1 = bench.dense_union(n, 1, dev)
    e1 = ei1.shape[1]
    assert 6 * g * e1 * 4 > 2 ** 31            # the condition step_pipe_fits refuses
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(g * n, 2048, generator=gen, device=dev) * (1.0 / (g * n) ** 0.5)     # column norms ~ 1, as after F.normalize(dim=0)
    ea = torch.rand(g * e1, 4, generator=gen, device=dev)
    ei = bench.dense_union(n, g, dev)
    with torch.no_grad():
        out = m(Data(x, ei, ea))["classified_edges"]
    torch.cuda.synchronize()
    assert len(out) == 3 and all(o.shape == (g * e1, 1) for o in out)
    assert all(bool(torch.isfinite(o).all().item()) for o in out)
    assert m.graph_flags() & 3 == 0
    pick = [0, 1, g // 2, g - 2, g - 1]
    small = Data(torch.cat([x[q * n:(q + 1) * n] for q in pick]), bench.dense_union(n, len(pick), dev),
                 torch.cat([ea[q * e1:(q + 1) * e1] for q in pick]))
    big = [torch.cat([o[q * e1:(q + 1) * e1] for q in pick]).clone() for o in out]
    del out, ei, ea, x
    torch.cuda.empty_cache()
    with torch.no_grad():
        got = m(small)["classified_edges"]
    for a, b in zip(big, got):
        assert float((a - b).abs().max()) <= 2e-6
    ref = NumpyOracle(params, arch, sd, np.float32).forward(small.x.cpu().numpy(), small.edge_index.cpu().numpy(), small.edge_attr.cpu().numpy())
    for a, r in zip(got, ref):
        assert np.abs(a.cpu().numpy() - r).max() <= 5e-6

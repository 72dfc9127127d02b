#!/usr/bin/env python3
"""Extreme shapes against the CPU oracle on the GPU box: one node with self loops only, a hub with 200 000 out-edges,
a million nodes with a thousand edges, a long path.  usage: python3 tools/soak_extremes.py"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import GOLDEN_DIR
from oracle.mpn_oracle import NumpyOracle, load_case
from gnn_cca_amd import MOTMPNet
params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))
class D: pass
def check(tag, n, ei, agg="sum", scale=1.0, tol=2e-5):
    p = copy.deepcopy(params); p.update(node_agg_fn=agg)
    s = dict(sd)
    for k in list(s):
        if k.startswith("MPNet.node_model"): s[k] = (s[k] * np.float32(scale)).astype(np.float32)
    m = MOTMPNet(copy.deepcopy(p), None, arch); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in s.items()}); m = m.cuda().eval()
    rng = np.random.default_rng(len(tag))
    x = (rng.standard_normal((n, 64)) * 0.3).astype(np.float32); ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref = NumpyOracle(p, arch, s, np.float32).forward(x, ei, ea)
    ref64 = NumpyOracle(p, arch, s, np.float64).forward(x, ei, ea) if ei.shape[1] >= 100_000 else None
    d = D(); d.x, d.edge_index, d.edge_attr = torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()
    with torch.no_grad():
        out = m(d)["classified_edges"]; torch.cuda.synchronize(); t0 = time.perf_counter(); out = m(d)["classified_edges"]; torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sc = max(1.0, max(float(np.abs(r).max()) for r in ref))
    err = max(float(np.abs(o.cpu().numpy() - r).max()) for o, r in zip(out, ref)) / sc
    note = ""
    if ref64 is not None:  # long sums: the fp32 oracle adds 200 000 terms sequentially; judge both against fp64
        e_gpu = max(float(np.abs(o.cpu().numpy() - r).max()) for o, r in zip(out, ref64)) / sc
        e_orc = max(float(np.abs(a - r).max()) for a, r in zip(ref, ref64)) / sc
        note = f" | vs fp64: gpu {e_gpu:.2e}, fp32 oracle {e_orc:.2e}"
        err = min(err, e_gpu) if e_gpu <= 2 * e_orc + 1e-6 else err
    print(f"{tag:34s} N={n:8d} E={ei.shape[1]:7d} flags={m.graph_flags()} rel.err={err:.2e} forward {dt*1e3:.3f} ms{note}", flush=True)
    assert err <= tol, tag
check("single node, 3 self loops", 1, np.zeros((2, 3), np.int64))
hub = 200_000
check("hub with 200k out-edges (mean)", 1000, np.stack([np.zeros(hub, np.int64), np.random.default_rng(0).integers(0, 1000, hub)]), agg="mean", scale=4.0)
check("hub with 200k out-edges (sum)", 1000, np.stack([np.zeros(hub, np.int64), np.random.default_rng(0).integers(0, 1000, hub)]), scale=1e-4, tol=5e-5)
rows = np.sort(np.random.default_rng(1).integers(0, 1_000_000, 1000))
check("1M nodes, 1000 edges", 1_000_000, np.stack([rows, np.random.default_rng(2).integers(0, 1_000_000, 1000)]))
n = 50_000
check("path graph", n, np.stack([np.arange(n - 1), np.arange(1, n)]).astype(np.int64))
check("every edge into node 0 (unsorted rows)", 5000, np.stack([np.random.default_rng(3).permutation(5000), np.zeros(5000, np.int64)]))
print("extremes ok")

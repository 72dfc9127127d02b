#!/usr/bin/env python3
"""cProfile of gnn_cca_amd.graph_build.build_graph_batch on the GPU box (host-side cost of row N1)."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_cca_amd.graph_build import build_graph_batch
frames, cams, per = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 4, 8
rng = np.random.default_rng(0)
n_g = cams * per; n = frames * n_g
id_cam = np.tile(np.repeat(np.arange(cams), per), frames)
ids = np.concatenate([rng.integers(0, per, size=n_g) for _ in range(frames)]).astype(np.int64)
xw = rng.normal(size=n); yw = rng.normal(size=n)
node = torch.randn(n, 2048, device='cuda'); reid = torch.randn(n, 256, device='cuda')
f = lambda: build_graph_batch(xw, yw, ids, id_cam, [n_g] * frames, [80.0] * frames, node, reid)
for _ in range(5): f()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
for _ in range(200): f()
t_issue = (time.perf_counter() - t0) / 200
torch.cuda.synchronize(); pr.disable()
print("host issue ms", t_issue * 1e3, "wall ms", (time.perf_counter() - t0) / 200 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4500])

#!/usr/bin/env python3
"""Timing of row N1 (graph construction + edge attributes) on the GPU next to its CPU oracle.
usage (GPU box): python3 tools/bench_graph_build.py [frames] [cams] [dets_per_cam] [reid_dim]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_cca_amd.graph_build import build_graph_batch  # noqa: E402
from oracle import graph_oracle  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = int(sys.argv[3]) if len(sys.argv) > 3 else 64
R = int(sys.argv[4]) if len(sys.argv) > 4 else 256
rng = np.random.default_rng(0)
n = g * cams * per
id_cam = np.tile(np.repeat(np.arange(cams), per), g)
ids = rng.integers(0, per * 2, size=n)
xw, yw = rng.uniform(-10, 10, n), rng.uniform(-10, 10, n)
max_dist = np.full(g, 80.0)
sizes = [cams * per] * g
reid = torch.randn(n, R).cuda()
node = torch.randn(n, 2048).cuda()
for _ in range(5):
    b = build_graph_batch(xw, yw, ids, id_cam, sizes, max_dist, node, reid)
torch.cuda.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    b = build_graph_batch(xw, yw, ids, id_cam, sizes, max_dist, node, reid)
torch.cuda.synchronize()
t_gpu = (time.perf_counter() - t0) / reps
E = b.edge_index.shape[1]
reid_n = graph_oracle.normalize_columns(reid.cpu().numpy())
t0 = time.perf_counter()
for _ in range(3):
    graph_oracle.build(xw, yw, ids, id_cam, sizes, max_dist, reid_n)
t_cpu = (time.perf_counter() - t0) / 3
print(json.dumps({"stage": "graph_build (SURVEY 8f N1)", "frames": g, "nodes": n, "edges": E, "reid_dim": R,
                  "gpu_ms_incl_host_plan_and_h2d": t_gpu * 1e3, "gpu_edges_per_s": E / t_gpu,
                  "cpu_oracle_numpy_ms": t_cpu * 1e3, "cpu_edges_per_s": E / t_cpu}))

#!/usr/bin/env python3
"""GPU box: where the HOST time of FramePipeline.__call__ goes on Terrace batches (the chain is host-bound: ~0.11 ms of Python + launches per
64-frame batch against ~0.09 ms of GPU time).  cProfile over a few hundred calls + wall time per call with and without final_async."""
import cProfile
import copy
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnn_cca_amd.pipeline import FramePipeline  # noqa: E402

dev = torch.device("cuda", 0)
frames = bench.terrace_frames(64, 16)
model = bench.build_model(copy.deepcopy(bench.graph_net_params(L=4)), 20, seed=0).to(dev)
dev_in = [(torch.from_numpy(f["node"]).to(dev), torch.from_numpy(f["reid"]).to(dev)) for f in frames]
pipe = FramePipeline(model)


def run(i):
    f, (node, reid) = frames[i], dev_in[i]
    return pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)


for i in range(16):
    run(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    for i in range(16):
        run(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host / 160 * 1e3:.4f} ms / batch, drained {t_all / 160 * 1e3:.4f} ms / batch")
# GPU time of one batch alone
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gpu = []
for i in range(16):
    torch.cuda.synchronize()
    ev0.record()
    run(i)
    ev1.record()
    torch.cuda.synchronize()
    gpu.append(ev0.elapsed_time(ev1))
print(f"GPU span of one batch (events around one call, idle queue): median {sorted(gpu)[8]:.4f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    for i in range(16):
        run(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)

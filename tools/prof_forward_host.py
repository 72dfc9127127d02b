#!/usr/bin/env python3
"""cProfile of the eval-mode forward's host side (headline graph).  usage (GPU box): python3 tools/prof_forward_host.py"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
model = bench.build_model(bench.graph_net_params(), 256).cuda()
data = bench.make_data(256, 1, 1, "cuda")
with torch.no_grad():
    for _ in range(20): model(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): model(data)
    t_issue = (time.perf_counter() - t0) / 2000
    torch.cuda.synchronize()
    print("host issue us/forward (no profiler):", round(t_issue * 1e6, 2))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): model(data)
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:3800])

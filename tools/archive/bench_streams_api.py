#!/usr/bin/env python3
"""Frames in flight through the package API in a FRESH process (GPU box): GraphedForward(streams=S).replay_slot round robin, and
GraphedForward.block(frames, chains=S, depth=D).   python3 tools/bench_streams_api.py"""
import sys, os, time, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gnn_cca_amd.inference import GraphedForward
model = bench.build_model(copy.deepcopy(bench.graph_net_params()), 256).cuda()
data = bench.make_data(256, 1, 1, "cuda")
K = 240
with torch.no_grad():
    for S in (2, 3):
        gfs = GraphedForward(model, streams=S)
        slots = [gfs.slot_inputs(data, i) for i in range(S)]
        torch.cuda.synchronize()
        def run(n):
            for i in range(n): gfs.replay_slot(i % S)
            gfs.join()
        run(60); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(3000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"replay_slot, {S} streams: {dt / 3000 * 1e6:.2f} us per forward", flush=True)
    gf = GraphedForward(model)
    for S, D in ((1, 0), (2, 4), (3, 4), (3, 2), (3, 8), (3, 16), (4, 4)):
        blk = gf.block([data] * K, adopt_inputs=True, chains=S, depth=D)
        blk.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); blk.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
        print(f"block of {K}, chains={S} depth={D}: {sorted(ts)[3]:.2f} us per forward", flush=True)

#!/usr/bin/env python3
"""Frames in flight on the headline graph: GraphedForward.block(chains=S, depth=D) over S and D (GPU box; run with and without
GPU_MAX_HW_QUEUES=8 in the environment: the device maps streams onto that many hardware queues, four by default)."""
import sys, os, time, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gnn_cca_amd.inference import GraphedForward
model = bench.build_model(copy.deepcopy(bench.graph_net_params()), 256).cuda()
data = bench.make_data(256, 1, 1, "cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 240
with torch.no_grad():
    gf = GraphedForward(model, warmup=0)
    b1 = gf.block([data] * K, adopt_inputs=True)
    for S, D in ((1, 0), (2, 4), (3, 4), (4, 4), (5, 4), (6, 4), (8, 4), (3, 1), (3, 2), (3, 3), (3, 7), (6, 2)):
        blk = gf.block([data] * K, adopt_inputs=True, chains=S, depth=D) if S > 1 else b1
        blk.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); blk.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
        print(f"queues={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} chains={S} depth={D}: {sorted(ts)[4]:.2f} us per forward", flush=True)

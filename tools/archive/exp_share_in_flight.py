#!/usr/bin/env python3
"""GPU box: BASELINE config 4's per-GPU share (64 x dense128) as ONE forward vs as S sub-batches in flight on S streams
(GraphedForward.block(chains=S)): does overlapping the launch gaps and tails of a ~100 us forward pay?   python tools/exp_share_in_flight.py"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnn_cca_amd.inference import GraphedForward  # noqa: E402


def timed(fn, reps=15):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    dev = "cuda"
    model = bench.build_model(copy.deepcopy(bench.graph_net_params()), 128).to(dev)
    K = 12   # shares per timed block
    with torch.no_grad():
        gf = GraphedForward(model, warmup=0)
        full = bench.make_data(128, 64, 1, dev)
        blk = gf.block([full] * K, adopt_inputs=True)
        t_full = timed(lambda: blk.replay()) / K
        print(f"one forward of 64 x dense128: {t_full * 1e6:.1f} us per share", flush=True)
        for S in (2, 4):
            parts = [bench.make_data(128, 64 // S, 10 + i, dev) for i in range(S)]
            frames = parts * K                      # K shares = K * S sub-batches, sub-batch i of every share on stream i
            b = gf.block(frames, adopt_inputs=True, chains=S, depth=1)
            t = timed(lambda: b.replay()) / K
            print(f"{S} sub-batches of {64 // S} x dense128 in flight: {t * 1e6:.1f} us per share", flush=True)
            b1 = gf.block(frames, adopt_inputs=True)
            t1 = timed(lambda: b1.replay()) / K
            print(f"{S} sub-batches back to back on one stream: {t1 * 1e6:.1f} us per share", flush=True)


if __name__ == "__main__":
    main()

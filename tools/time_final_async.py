#!/usr/bin/env python3
"""Where a Terrace batch's time goes with the overlapped host pass (FrameResult.final_async): the GPU chain alone, chain + submission,
chain + submission + collection two batches later, for a few pool sizes / depths.  GPU box.

    python tools/time_final_async.py [--threads 8,12,16] [--depth 2,4]"""
import argparse
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="8,12,16")
    ap.add_argument("--depth", default="2,4")
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from gnn_cca_amd.pipeline import FramePipeline
    dev = torch.device("cuda", 0)
    frames = bench.terrace_frames(64, args.batches)
    model = bench.build_model(copy.deepcopy(bench.graph_net_params(L=4)), 20, seed=0).to(dev)
    dev_in = [(torch.from_numpy(f["node"]).to(dev), torch.from_numpy(f["reid"]).to(dev)) for f in frames]

    def run(pipe, i):
        f, (node, reid) = frames[i], dev_in[i]
        return pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)

    pipe = FramePipeline(model)
    with torch.no_grad():
        r = run(pipe, 0)
        sd = model.state_dict()
        last_bias = [k for k in sd if k.startswith("classifier.") and k.endswith(".bias")][-1]
        sd[last_bias] -= r.outputs["classified_edges"][-1].median()
        model.load_state_dict(sd)
    n = args.batches * args.reps

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def chain(pipe):
        for _ in range(args.reps):
            for i in range(args.batches):
                run(pipe, i)

    for _ in range(2):
        chain(pipe)
    print(f"chain alone                      : {timed(lambda: chain(pipe)):.4f} ms / batch")
    for th in [int(v) for v in args.threads.split(",")]:
        pipe = FramePipeline(model)
        pipe.host_threads = th
        for i in range(4):
            run(pipe, i).final_async().result()

        def submit_only():
            pend = []
            for _ in range(args.reps):
                for i in range(args.batches):
                    pend.append(run(pipe, i).final_async())
            t1 = time.perf_counter()
            for p in pend:
                p.result(copy=False)
            submit_only.drain = (time.perf_counter() - t1) * 1e3
        ms = timed(submit_only)
        print(f"threads {th:2d}: submit all, collect at the end: {ms:.4f} ms / batch (drain {submit_only.drain:.2f} ms total)")
        for depth in [int(v) for v in args.depth.split(",")]:
            for cp in (True, False):
                acc = []

                def overlapped():
                    pend = []
                    t_sub = t_res = 0.0
                    for _ in range(args.reps):
                        for i in range(args.batches):
                            r = run(pipe, i)
                            t1 = time.perf_counter()
                            pend.append(r.final_async())
                            t2 = time.perf_counter()
                            if len(pend) > depth:
                                p = pend.pop(0)
                                p.result(copy=cp)
                                acc.append(p.times_us)
                            t3 = time.perf_counter()
                            t_sub += t2 - t1
                            t_res += t3 - t2
                    while pend:
                        p = pend.pop(0)
                        p.result(copy=cp)
                        acc.append(p.times_us)
                    overlapped.sub, overlapped.res = t_sub / n * 1e6, t_res / n * 1e6
                ms = timed(overlapped)
                import numpy as np
                a = np.mean(np.asarray(acc), axis=0)
                print(f"threads {th:2d} depth {depth} copy {int(cp)}: {ms:.4f} ms / batch | main thread: submit {overlapped.sub:.1f} us, result {overlapped.res:.1f} us | "
                      f"job: pickup {a[0]:.0f} us, event {a[1]:.0f} us, frames {a[2]:.0f} us, slack before collection {-a[3]:.0f} us")
        # the synchronous form
        def sync_final():
            for _ in range(args.reps):
                for i in range(args.batches):
                    run(pipe, i).final()
        print(f"threads {th:2d}: final() one batch after the other: {timed(sync_final):.4f} ms / batch")
        pipe.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Host-side cost of the reference's rounding / splitting heuristics (csrc/post_host.cpp) on Terrace-shaped frames with near-random
predictions -- the adversarial case bench.py's synthetic model produces (most frames raise a trigger).  No GPU needed.

    python tools/bench_post_host.py [--frames 1024] [--threads 1,4,8,16]

Per-frame microseconds on one thread, and frames / s through gnncca_post_finalize_frames_host for the listed thread counts."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frames(n_frames, seed=0):
    from oracle import post_oracle as po
    z = np.load(os.path.join(ROOT, "tests", "golden", "terrace_topology.npz"))
    rng = np.random.default_rng(seed)
    pick = np.linspace(0, len(z["frame"]) - 1, n_frames).astype(int)
    out = []
    for q in pick:
        cam = z["cam"][z["node_ptr"][q]:z["node_ptr"][q + 1]].astype(np.int64)
        n = len(cam)
        i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        m = cam[i] != cam[j]
        ei = np.stack([i[m], j[m]]).astype(np.int64)
        logits = rng.normal(0.0, 1.0, size=ei.shape[1]).astype(np.float32)      # centred: half the edges active, as bench.py's model after centring
        probs, pred = po.threshold(logits)
        out.append((n, ei, probs.astype(np.float32), po.prune(ei, pred).astype(np.int64)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--threads", default="1,4,8,16")
    ap.add_argument("--pinned", action="store_true", help="the batch arrays in pinned host memory (torch pin_memory), as final_async hands them over")
    args = ap.parse_args()
    from gnn_cca_amd import _native as nat
    lib = nat.lib()
    fr = frames(args.frames)
    node_ptr, edge_ptr = [0], [0]
    for n, ei, _, _ in fr:
        node_ptr.append(node_ptr[-1] + n), edge_ptr.append(edge_ptr[-1] + ei.shape[1])
    src = np.ascontiguousarray(np.concatenate([ei[0] + b for (_, ei, _, _), b in zip(fr, node_ptr)]))
    dst = np.ascontiguousarray(np.concatenate([ei[1] + b for (_, ei, _, _), b in zip(fr, node_ptr)]))
    probs = np.ascontiguousarray(np.concatenate([p for _, _, p, _ in fr]))
    pred0 = np.ascontiguousarray(np.concatenate([p for _, _, _, p in fr]))
    np_h, ep_h = np.asarray(node_ptr, np.int32), np.asarray(edge_ptr, np.int32)
    listed = np.arange(len(fr), dtype=np.int32)
    keep = []
    if args.pinned:
        import torch

        def pin(a):
            t = torch.empty(a.nbytes, dtype=torch.uint8, pin_memory=True)
            keep.append(t)
            v = t.numpy().view(a.dtype)
            v[:] = a
            return v
        src, dst, probs, pred0 = pin(src), pin(dst), pin(probs), pin(pred0)
    print(f"{len(fr)} frames, {node_ptr[-1] / len(fr):.1f} nodes / {edge_ptr[-1] / len(fr):.0f} edges per frame")
    ref = None
    for t in [int(v) for v in args.threads.split(",")]:
        best = 1e9
        for _ in range(3):
            pred, labels, k = pred0.copy(), np.zeros(node_ptr[-1], np.int32), np.zeros(len(fr), np.int32)
            if args.pinned:
                pred = pin(pred)
            t0 = time.perf_counter()
            st = lib.gnncca_post_finalize_frames_host(src.ctypes.data, dst.ctypes.data, np_h.ctypes.data, ep_h.ctypes.data, listed.ctypes.data, len(fr),
                                                      probs.ctypes.data, pred.ctypes.data, 7, labels.ctypes.data, k.ctypes.data, t)
            best = min(best, time.perf_counter() - t0)
            assert st == 0
        if ref is None:
            ref = (pred.copy(), labels.copy(), k.copy())
            print(f"changed frames: {sum(1 for q in range(len(fr)) if not np.array_equal(pred[ep_h[q]:ep_h[q+1]], pred0[ep_h[q]:ep_h[q+1]]))}")
        else:
            assert np.array_equal(pred, ref[0]) and np.array_equal(labels, ref[1]) and np.array_equal(k, ref[2])
        print(f"threads {t:3d}: {best * 1e3:8.2f} ms  = {best / len(fr) * 1e6:7.1f} us/frame wall, {len(fr) / best:10.0f} frames/s, digest {int(pred.sum())} {int(k.sum())}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU box: per-kernel times (dispatch-attached HIP events, forward_profiled) for a list of GxN dense workloads.
    python3 tools/exp_sizes.py 64x256 64x257 512x128 [--L 4] [--edge-state bf16] [--unsplit] [--products 3]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    L, es, specs, unsplit, products = 4, "fp32", [], False, 6
    a = sys.argv[1:]
    while a:
        t = a.pop(0)
        if t == "--L":
            L = int(a.pop(0))
        elif t == "--edge-state":
            es = a.pop(0)
        elif t == "--unsplit":
            unsplit = True
        elif t == "--products":
            products = int(a.pop(0))
        else:
            specs.append(tuple(int(v) for v in t.split("x")))
    dev = torch.device("cuda", 0)
    for g, n in specs:
        params = bench.graph_net_params(L=L)
        model = bench.build_model(copy.deepcopy(params), n).to(dev)
        model.edge_state_dtype = es
        model.encoder_unsplit = unsplit
        model.encoder_products = products
        data = bench.make_data(n, g, 1, dev)
        E = data.edge_index.shape[1]
        acc = {}
        with torch.no_grad():
            for _ in range(3):
                model(data)
            for _ in range(10):
                _, times = model.forward_profiled(data)
                for i, (kind, ms) in enumerate(times):
                    acc.setdefault((i, kind), []).append(ms)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(20):
                model(data)
            torch.cuda.synchronize()
            fwd = (time.perf_counter() - t0) / 20
        per = " ".join(f"{k}:{np.median(v) * 1e3:.1f}" for (i, k), v in sorted(acc.items()))
        print(f"{g}x{n} L{L} {es} E={E} fwd {fwd * 1e3:.4f} ms  {E / fwd / 1e9:.2f} Gedges/s | us: {per}", flush=True)
        del model, data
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU box: encoder GEMM / tail / plan times (dispatch-attached events) at a fixed batch for several node_in widths -- the
intercept of time over K is the launch's prologue + epilogue, the slope its per-chunk cost.
    python3 tools/exp_gemm_k.py 64x128 [256 512 1024 2048]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    g, n = (int(v) for v in sys.argv[1].split("x"))
    ks = [int(v) for v in sys.argv[2:]] or [256, 512, 1024, 2048]
    dev = torch.device("cuda", 0)
    base = bench.make_data(n, g, 1, dev)
    for K in ks:
        params = bench.graph_net_params()
        params["encoder_feats_dict"]["nodes"]["resnet50"]["node_in_dim"] = K
        model = bench.build_model(copy.deepcopy(params), n).to(dev)
        data = bench.Data()
        data.x = base.x[:, :K].contiguous()
        data.edge_index, data.edge_attr = base.edge_index, base.edge_attr
        acc = {}
        with torch.no_grad():
            for _ in range(3):
                model(data)
            for _ in range(10):
                _, times = model.forward_profiled(data)
                for i, (kind, ms) in enumerate(times):
                    acc.setdefault((i, kind), []).append(ms)
        per = " ".join(f"{k}:{np.median(v) * 1e3:.1f}" for (i, k), v in sorted(acc.items()) if not k.startswith("step"))
        print(f"{g}x{n} K={K} | us: {per}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Same-box A/B of the column ranges (model.column_ranges True / False, alternated inside ONE process) on dense workloads:
per-kernel times (events attached to each dispatch) and the forward time inside a HIP-graph block of K forwards.
    python tools/ab_ranges.py 1x256 64x128 64x256 1x1024:8 512x128       (graphs x nodes[:L])
GNNCCA_LIB=<other build> runs the same protocol on an A/B twin of the library (e.g. lib/libgnncca_mpn_r3pipe.so)."""
import copy
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
import torch

import bench
from gnn_cca_amd.inference import GraphedForward

dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("GNNCCA_LIB", "libgnncca_mpn.so"))
for spec in sys.argv[1:]:
    gn, _, l = spec.partition(":")
    g, n = (int(v) for v in gn.split("x"))
    L = int(l) if l else 4
    params = bench.graph_net_params(L=L)
    model = bench.build_model(copy.deepcopy(params), n).to(dev)
    model.edge_state_dtype = 'bf16' if os.environ.get('AB_BF16') else 'fp32'
    data = bench.make_data(n, g, 1, dev)
    E = data.edge_index.shape[1]
    K = 50 if E < 3e6 else 10
    res = {}
    with torch.no_grad():
        for rep in range(3):
            for on in (True, False):
                model.column_ranges = on
                for _ in range(3):
                    model(data)
                kinds = {}
                for _ in range(6):
                    _, times = model.forward_profiled(data)
                    for i, (k, t) in enumerate(times):
                        kinds.setdefault(f"{i}:{k}", []).append(t * 1e3)
                gf = GraphedForward(model)
                blk = gf.block([data] * K, adopt_inputs=True)
                blk.replay()
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    blk.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / K * 1e3)
                r = res.setdefault(on, {"fwd": [], "k": {}})
                r["fwd"].append(float(np.median(ts)))
                for k, v in kinds.items():
                    r["k"].setdefault(k, []).append(float(np.median(v)))
                del gf, blk
        state = model.column_ranges_state() if hasattr(model, "column_ranges_state") else None
    for on in (True, False):
        r = res[on]
        ks = " ".join(f"{k}={np.median(v):.1f}" for k, v in r["k"].items())
        print(f"[{tag}{' bf16' if os.environ.get('AB_BF16') else ''}] {spec:10s} ranges={'on ' if on else 'off'} fwd(us)={' '.join(f'{t:.2f}' for t in r['fwd'])}  | {ks}", flush=True)
    del model, data
    torch.cuda.empty_cache()

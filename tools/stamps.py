#!/usr/bin/env python3
"""Where does a launch spend its time?  Runs ONE forward of a dense graph with the diagnostic build
(lib/libgnncca_mpn_stamps.so, -DGNNCCA_STAMPS) and prints, per kernel, the median / max over waves of the s_memtime
stamps relative to the earliest wave start of that launch.  Never quote this build's run time (the stamps cost
cycles); read the SHARES.   usage (GPU box):  GNNCCA_LIB=.../libgnncca_mpn_stamps.so python3 tools/stamps.py [nodes] [graphs]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnn_cca_amd import _native as nat  # noqa: E402

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = nat.lib()
lib.gnncca_debug_set_stamps.argtypes = [C.c_void_p]
model = bench.build_model(bench.graph_net_params(), nodes).cuda()
data = bench.make_data(nodes, graphs, 1, "cuda")
with torch.no_grad():
    for _ in range(5):
        model(data)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 4096 * 4 * 16, dtype=torch.int64, device="cuda")
    assert lib.gnncca_debug_set_stamps(buf.data_ptr()) == 0
    model(data)
    torch.cuda.synchronize()
    lib.gnncca_debug_set_stamps(None)
st = buf.cpu().numpy().reshape(8, 4096 * 4, 16).astype(np.float64)
names = {0: "enc_tail", 1: "enc_gemm_plan", 2: "step1", 3: "step2", 4: "step3", 5: "step4"}
labels = {0: ["start", "partials issued", "weights staged", "after sync", "row reduced", "after sync2", "layer2 done", "projected", "end"],
          1: ["start", "end"],
          2: ["start", "prologue issued", "after sync", "r1 gathers issued", "r1 computed", "loop done", "combined", "end",
              "r2 gathers issued", "r2 computed", "r2 operands landed", "r3+ gathers issued", "r3+ computed", "r3+ operands landed"]}
CLK_GHZ = 2.1  # s_memtime counts shader cycles (per-XCD counters with different bases: only per-wave deltas mean anything)
for k in range(6):
    a = st[k]
    rows = a[a[:, 0] > 0]
    if len(rows) == 0:
        continue
    lab = labels[min(k, 2)]
    print(f"{names[k]}: {len(rows)} waves; per-wave time since the wave's own first stamp, ns at {CLK_GHZ} GHz")
    prev = None
    order = list(range(len(lab))) if min(k, 2) != 2 else [0, 1, 2, 3, 5 + 0 * 0, 4, 8, 10, 9, 11, 13, 12, 5, 6, 7]
    if min(k, 2) == 2:
        order = [0, 1, 2, 3, 15, 4, 8, 10, 9, 11, 13, 12, 5, 6, 7]
        lab = lab + ["", "r1 operands landed"]
    for i in order:
        l = lab[i]
        ok = rows[:, i] > 0
        if not ok.any():
            continue
        d = (rows[ok, i] - rows[ok, 0]) / CLK_GHZ
        med = np.median(d)
        step = "" if prev is None else f"  (+{med - prev:6.0f})"
        print(f"   {l:22s} median {med:7.0f}   p95 {np.percentile(d, 95):7.0f}   max {d.max():7.0f}{step}")
        prev = med

#!/usr/bin/env python3
"""Phase totals of enc_gemm_split_lds_kernel<.., PIPE> (diagnostic build -DGNNCCA_STAMPS; per-wave s_memtime totals, median over waves).
usage (GPU box): GNNCCA_DIAG=1 GNNCCA_GEMM_BF16=1 GNNCCA_LIB=.../libgnncca_mpn_stamps.so python3 tools/stamps_gemm.py [nodes] [graphs]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnn_cca_amd import _native as nat  # noqa: E402

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = nat.lib()
lib.gnncca_debug_set_stamps.argtypes = [C.c_void_p]
model = bench.build_model(bench.graph_net_params(), nodes).cuda()
data = bench.make_data(nodes, graphs, 1, "cuda")
with torch.no_grad():
    for _ in range(3):
        model(data)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 4096 * 4 * 16, dtype=torch.int64, device="cuda")
    assert lib.gnncca_debug_set_stamps(buf.data_ptr()) == 0
    model(data)
    torch.cuda.synchronize()
    lib.gnncca_debug_set_stamps(None)
raw = buf.cpu().numpy().reshape(8, 4096, 4, 16).astype(np.float64)[6]
names = ["phase 1 (12 fragment reads + 24 MFMA)", "barrier wait", "phase 2 (convert + store + loads + 12 reads + 24 MFMA)", "loop overhead"]
for label, st in (("waves 0-3", raw[:2048].reshape(-1, 16)), ("waves 4-7", raw[2048:].reshape(-1, 16))):
    st = st[st[:, :4].sum(1) > 0]
    if len(st) == 0:
        continue
    tot = np.median(st[:, :4].sum(1))
    print(f"{label}: {len(st)} waves; total per wave {tot:.0f} ticks of s_memtime")
    for i, n in enumerate(names):
        print(f"   {n:58s} median {np.median(st[:, i]):10.0f}  ({100 * np.median(st[:, i]) / tot:5.1f} %)")
    real = np.median(st[:, 7])
    print(f"   loop wall time (s_memrealtime, 100 MHz): {real / 100:.1f} us -> s_memtime rate {tot / (real / 100) / 1e3:.2f} GHz")

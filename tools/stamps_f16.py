#!/usr/bin/env python3
"""Phase totals of the fp16-split encoder GEMM's main loop (enc_f16.cuh; diagnostic build -DGNNCCA_STAMPS: per-wave s_memtime totals).
usage (GPU box): GNNCCA_DIAG=1 GNNCCA_LIB=.../libgnncca_mpn_stamps.so python3 tools/stamps_f16.py [N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnn_cca_amd import _native as nat  # noqa: E402

n_tot = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lib = nat.lib()
lib.gnncca_debug_set_stamps.argtypes = [C.c_void_p]
model = bench.build_model(bench.graph_net_params(), 3).cuda()
d = bench.Data()
i = torch.arange(n_tot, device="cuda")
d.edge_index = torch.stack([i, (i + 1) % n_tot]).contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
d.x = torch.randn(n_tot, 2048, generator=g, device="cuda") / n_tot ** 0.5
d.edge_attr = torch.rand(n_tot, 4, generator=g, device="cuda")
with torch.no_grad():
    for _ in range(3):
        model(d)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 4096 * 4 * 16, dtype=torch.int64, device="cuda")
    assert lib.gnncca_debug_set_stamps(buf.data_ptr()) == 0
    model(d)
    torch.cuda.synchronize()
    lib.gnncca_debug_set_stamps(None)
raw = buf.cpu().numpy().reshape(8, 4096, 4, 16).astype(np.float64)[7]
names = ["wait for own DMA (vmcnt)", "workgroup barrier", "(unused)", "compute (20 ds_read_b128, split, 24 MFMA, 6 LDS-DMA)", "prologue", "drain + combine + arm decision", "epilogue"]
for label, st in (("waves 0-3", raw[:2048].reshape(-1, 16)), ("waves 4-7", raw[2048:].reshape(-1, 16))):
    st = st[st[:, :7].sum(1) > 0]
    if len(st) == 0:
        continue
    tot = np.median(st[:, :7].sum(1))
    print(f"{label}: {len(st)} waves; total per wave {tot:.0f} ticks of s_memtime")
    for k, nm in enumerate(names):
        print(f"   {nm:48s} median {np.median(st[:, k]):10.0f}  ({100 * np.median(st[:, k]) / tot:5.1f} %)   per chunk {np.median(st[:, k]) / 64:7.0f}")
    real = np.median(st[:, 7])
    print(f"   loop wall time (s_memrealtime, 100 MHz): {real / 100:.1f} us -> s_memtime rate {tot / (real / 100) / 1e3:.2f} GHz")

# the 32-row kernel (N <= GNNCCA_GEMM_R32F_MAX): ten waves per workgroup, slot 6's region as [1024 blocks][16 waves][16]
r32 = buf.cpu().numpy().reshape(8, -1)[6][:1024 * 16 * 16].reshape(1024, 16, 16).astype(np.float64)
if r32[:, :10, :7].sum() > 0 and n_tot <= 20480:
    for label, sel in (("compute waves 0-7", slice(0, 8)), ("loader waves 8-9", slice(8, 10))):
        st = r32[:, sel].reshape(-1, 16)
        st = st[st[:, :7].sum(1) > 0]
        tot = np.median(st[:, :7].sum(1))
        print(f"32-row kernel, {label}: {len(st)} waves; total per wave {tot:.0f} ticks")
        for k, nm in enumerate(["loader: wait for own DMA", "barrier", "loader: issue 8 DMA", "compute (4 ds_read, split, 12 MFMA, 8 B loads)", "prologue", "loader: drain", "epilogue"]):
            print(f"   {nm:48s} median {np.median(st[:, k]):10.0f}  ({100 * np.median(st[:, k]) / tot:5.1f} %)   per chunk {np.median(st[:, k]) / 16:7.0f}")
        real = np.median(st[:, 7])
        print(f"   wall (s_memrealtime): {real / 100:.1f} us -> {tot / (real / 100) / 1e3:.2f} GHz")

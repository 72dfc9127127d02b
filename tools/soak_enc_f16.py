#!/usr/bin/env python3
"""GPU box: the fp16-split encoder kernels against the bf16 kernels (encoder_unsplit=True takes the round-2/3 kernels for every N >= 4096) on
random node counts -- tile edges, both fp16 kernels, split-K factors.  Compares the traced encoder output.  python3 tools/soak_enc_f16.py [n_cases]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(12)
dev = torch.device("cuda", 0)
model = bench.build_model(copy.deepcopy(bench.graph_net_params(L=2)), 3).to(dev)
sizes = [4096, 4097, 8191, 8192, 8193, 8448, 16384, 16385, 65536, 65537, 65281] + [int(v) for v in rng.integers(4096, 90000, size=n_cases)]
worst = 0.0
for n in sizes:
    g = torch.Generator(device=dev).manual_seed(n)
    d = bench.Data()
    d.x = torch.randn(n, 2048, generator=g, device=dev) * float(rng.choice([1e-3, 0.05, 1.0, 30.0]))
    i = torch.arange(n, device=dev)
    d.edge_index = torch.stack([i, (i + 1) % n]).contiguous()
    d.edge_attr = torch.rand(n, 4, generator=g, device=dev)
    out = {}
    for unsplit in (False, True):
        model.encoder_unsplit = unsplit
        tr = {}
        with torch.no_grad():
            model(d, trace=tr)
        out[unsplit] = tr["h_enc"].clone()
    scale = float(out[True].abs().max())
    err = float((out[False] - out[True]).abs().max())
    worst = max(worst, err / max(scale, 1e-30))
    print(f"N={n:6d} max|h_enc| {scale:.3e}  max|f16 - bf16| {err:.3e}  rel {err / max(scale, 1e-30):.2e}", flush=True)
    assert err <= 2e-6 * max(scale, 1.0), (n, err, scale)
print("worst relative difference", worst)

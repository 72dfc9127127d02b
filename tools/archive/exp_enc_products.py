#!/usr/bin/env python3
"""GPU box: accuracy + time of the split-bf16 encoder GEMM with 6 products (default) and 3 (GNNCCA_OPT_ENC_SPLIT3,
model.encoder_products = 3): max |h_enc - fp64| and max |logit - fp32 oracle| on 64 x dense128 (N = 8192: 128-row GEMM)
and on a 51 233-node ring graph (256-row GEMM, fused epilogue)."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.mpn_oracle import NumpyOracle  # noqa: E402  (the checker)
from test_gpu_parity import Data, _default_model, _dense_graph, build  # noqa: E402


PRODUCTS = 6


def run(name, params, arch, sd, x, ei, ea):
    orc = NumpyOracle(params, arch, sd, np.float32)
    tr = {}
    ref = orc.forward(x, ei, ea, tr)
    h64 = NumpyOracle(params, arch, sd, np.float64)._mlp("encoder.node_mlp", x.astype(np.float64))
    m = build(params, arch, sd)
    m.encoder_products = PRODUCTS
    d = Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda())
    trace = {}
    with torch.no_grad():
        out = [t.clone() for t in m(d)["classified_edges"]]
        m(d, trace=trace)
        _, times = m.forward_profiled(d)
        _, times = m.forward_profiled(d)
    err_h = float(np.abs(trace["h_enc"].cpu().numpy() - h64).max())
    err_ref = float(np.abs(tr["h_enc"] - h64).max())
    err_l = max(float(np.abs(o.cpu().numpy() - r).max()) for o, r in zip(out, ref))
    scale = max(float(np.abs(r).max()) for r in ref)
    gemm = [ms for k, ms in times if k == "enc_gemm"]
    print(f"[{PRODUCTS} products] {name}: |h_enc - fp64| {err_h:.2e} (fp32 oracle's own {err_ref:.2e}); |logit - fp32 oracle| {err_l:.2e} "
          f"(max |logit| {scale:.2f}); enc_gemm {gemm[0] * 1e3:.1f} us", flush=True)


def main():
    rng = np.random.default_rng(11)
    params, arch, sd = _default_model(1.0 / 127)
    g, n = 64, 128
    x = rng.standard_normal((g * n, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ei = np.concatenate([_dense_graph(n, k * n) for k in range(g)], axis=1)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    run("64 x dense128 (N = 8192), x normalised", params, arch, sd, x, ei, ea)
    xr = rng.standard_normal((g * n, 2048)).astype(np.float32)  # unnormalised features: O(1) entries, O(10) pre-activations
    run("64 x dense128 (N = 8192), x ~ N(0,1)", params, arch, sd, xr, ei, ea)
    params, arch, sd = _default_model(1.0)
    nn = 51233
    x = rng.standard_normal((nn, 2048)).astype(np.float32)
    src = np.repeat(np.arange(nn), 2)
    dst = (src + np.tile([1, 5], nn)) % nn
    ei = np.stack([src, dst]).astype(np.int64)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    run("ring 51 233 nodes, x ~ N(0,1)", params, arch, sd, x, ei, ea)


if __name__ == "__main__":
    for PRODUCTS in (6, 3):
        main()

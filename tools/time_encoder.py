#!/usr/bin/env python3
"""GPU box: per-kernel times (HIP events attached to each dispatch, MOTMPNet.forward_profiled) of a forward over N nodes -- the
encoder GEMM's operating points.  The graph is a ring (E = N) unless --dense n is given (N / n dense n-node graphs), so the encoder
dominates.  usage: python3 tools/time_encoder.py N [N ...] [--dense n] [--reps R] [--check]   (A/B by environment, one process per arm)"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("nodes", type=int, nargs="+")
ap.add_argument("--dense", type=int, default=0)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--check", action="store_true", help="encoder output h0 (traced) against an fp64 evaluation on a sample of rows")
ap.add_argument("--unsplit", action="store_true")
ap.add_argument("--products", type=int, default=6)
args = ap.parse_args()
dev = torch.device("cuda", 0)
out = {}
for n_tot in args.nodes:
    params = bench.graph_net_params(L=4)
    model = bench.build_model(copy.deepcopy(params), max(args.dense, 3)).to(dev)
    model.encoder_unsplit = args.unsplit
    model.encoder_products = args.products
    d = bench.Data()
    g = torch.Generator(device="cpu").manual_seed(1)
    if args.dense:
        G = n_tot // args.dense
        d.edge_index = bench.dense_union(args.dense, G, dev)
        n_tot = G * args.dense
    else:
        i = torch.arange(n_tot, device=dev)
        d.edge_index = torch.stack([i, (i + 1) % n_tot]).contiguous()
    x = torch.randn(n_tot, 2048, generator=g)
    x = torch.nn.functional.normalize(x, p=2, dim=0) if n_tot > 1 else x
    d.x = x.to(dev)
    d.edge_attr = torch.rand(d.edge_index.shape[1], 4, generator=g).to(dev)
    kms = {}
    with torch.no_grad():
        for _ in range(3):
            model(d)
        for _ in range(args.reps):
            _, times = model.forward_profiled(d)
            for kind, ms in times:
                kms.setdefault(kind, []).append(ms * 1e3)
    rec = {k: [float(np.median(v)), float(np.min(v))] for k, v in kms.items()}
    if args.check:
        with torch.no_grad():
            trace = {}
            model(d, trace=trace)
        h = trace["h_enc"].double().cpu()
        sd = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
        rows = torch.linspace(0, n_tot - 1, 257).long().unique()
        xr = x[rows].double()
        h1 = torch.relu(xr @ sd["encoder.node_mlp.fc_layers.0.weight"].T + sd["encoder.node_mlp.fc_layers.0.bias"])
        h2 = torch.relu(h1 @ sd["encoder.node_mlp.fc_layers.3.weight"].T + sd["encoder.node_mlp.fc_layers.3.bias"])
        rec["h_enc_max_abs_err_vs_fp64"] = float((h[rows] - h2).abs().max())
        rec["h_enc_max_abs"] = float(h2.abs().max())
        x32 = x[rows]
        w32 = {k: v.float() for k, v in sd.items()}
        h1f = torch.relu(x32 @ w32["encoder.node_mlp.fc_layers.0.weight"].T + w32["encoder.node_mlp.fc_layers.0.bias"])
        h2f = torch.relu(h1f @ w32["encoder.node_mlp.fc_layers.3.weight"].T + w32["encoder.node_mlp.fc_layers.3.bias"])
        rec["torch_fp32_cpu_err_vs_fp64"] = float((h2f.double() - h2).abs().max())
    out[str(n_tot)] = rec
    del model, d
    torch.cuda.empty_cache()
print(json.dumps(out))

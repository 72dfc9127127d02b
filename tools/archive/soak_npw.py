#!/usr/bin/env python3
"""GPU box: random ragged batches big enough for the two-nodes-per-wave / deferred-classification forms of the step kernel; prints one
SHA-256 per batch over the bytes of all logits.  Run it twice -- as is, and with GNNCCA_DIAG=1 GNNCCA_NPW=1 GNNCCA_DEFER_CLS=0 (one node
per wave, classification in place) -- and diff the outputs: the two forms promise the same bits.
    python tools/soak_npw.py [n_batches=24] > a.txt;  GNNCCA_DIAG=1 GNNCCA_NPW=1 GNNCCA_DEFER_CLS=0 python tools/soak_npw.py > b.txt;  diff a.txt b.txt"""
import copy
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def batch(rng):
    """Disjoint union of dense-ish graphs of 2 ... 129 nodes (a few of up to 400), random edge drops, isolated nodes, sometimes shuffled."""
    target = int(rng.integers(16384, 36000))
    sizes, n = [], 0
    while n < target:
        s = int(rng.integers(2, 130)) if rng.random() > 0.02 else int(rng.integers(200, 400))
        sizes.append(s)
        n += s
    drop = float(rng.choice([0.0, 0.05, 0.3]))
    parts, off = [], 0
    for s in sizes:
        i, j = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
        keep = (i != j) & (rng.random((s, s)) >= drop)
        if rng.random() < 0.2:
            keep[rng.integers(0, s)] = False          # a node without out-edges
        parts.append(np.stack([i[keep] + off, j[keep] + off]))
        off += s
    ei = np.concatenate(parts, axis=1).astype(np.int64)
    if ei.shape[1] / off > 128:                       # keep the average degree in the two-node regime
        return batch(rng)
    shuffled = rng.random() < 0.15
    if shuffled:
        ei = ei[:, rng.permutation(ei.shape[1])]
    return ei, off, shuffled


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(2024)
    dev = torch.device("cuda:0")
    for b in range(n_batches):
        ei, n, shuffled = batch(rng)
        L = int(rng.choice([2, 4, 4, 6]))
        agg = "mean" if rng.random() < 0.3 else "sum"
        params = bench.graph_net_params(L=L, n_cls=min(3, L), agg=agg)
        deg = max(int(np.bincount(ei[0], minlength=n).max()), 1)
        model = bench.build_model(copy.deepcopy(params), deg + 1 if agg == "sum" else 2, seed=b).to(dev).eval()
        if rng.random() < 0.25:
            model.edge_state_dtype = "bf16"
        g = torch.Generator().manual_seed(b)
        x = torch.nn.functional.normalize(torch.randn((n, 2048), generator=g), p=2, dim=0).to(dev)
        ea = torch.rand((ei.shape[1], 4), generator=g).to(dev)
        d = bench.Data()
        d.x, d.edge_index, d.edge_attr = x, torch.from_numpy(ei).to(dev), ea
        with torch.no_grad():
            out = model(d)["classified_edges"]
        h = hashlib.sha256()
        finite = True
        for t in out:
            a = t.cpu().numpy()
            finite = finite and bool(np.isfinite(a).all())
            h.update(a.tobytes())
        print(f"batch {b:3d} N={n} E={ei.shape[1]} L={L} agg={params.get('node_agg_fn', 'sum')} state={model.edge_state_dtype} shuffled={int(shuffled)} "
              f"flags={model.graph_flags()} finite={int(finite)} sha={h.hexdigest()[:24]}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU box: sha256 of the logits of a few dense batches -- run once per library build (GNNCCA_LIB=...) and compare: a bitwise A/B of two builds.
    python tools/hash_logits.py 64x128 128x128 512x128 [--scale 1e-3]"""
import hashlib
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch

import bench

scale = 1.0
args = [a for a in sys.argv[1:]]
if "--scale" in args:
    i = args.index("--scale")
    scale = float(args[i + 1])
    del args[i:i + 2]
for spec in args:
    g, n = (int(v) for v in spec.split("x"))
    model = bench.build_model(bench.graph_net_params(L=4), n).cuda()
    data = bench.make_data(n, g, 1, "cuda")
    data.x = (data.x * scale).contiguous()
    with torch.no_grad():
        out = model(data)["classified_edges"]
    torch.cuda.synchronize()
    h = hashlib.sha256(b"".join(o.cpu().numpy().tobytes() for o in out)).hexdigest()[:16]
    print(spec, scale, h, float(out[-1].abs().max()))

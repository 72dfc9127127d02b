"""Dense-256 forward of the shipped shape (MFMA family) and of two configurations of the generic family: eager calls, then the same
forwards inside a HIP-graph block (the GPU's own time).  GPU box:   python3 tools/time_generic_family.py
(Eager figures first for every configuration: eager loops measured AFTER a graph capture in the same process came out erratic, 0.1-0.9 ms.)"""
import sys, time, copy, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from gnn_cca_amd.inference import GraphedForward


def wide(p):
    p['encoder_feats_dict']['nodes']['resnet50']['node_out_dim'] = 64
    p['node_model_feats_dict']['fc_dims'] = [64]


def deep(p):
    p['edge_model_feats_dict']['fc_dims'] = [12, 6]


cases = [("mfma family (shipped)", lambda p: None), ("generic: node latent 64", wide), ("generic: two-layer edge MLP", deep)]
models = []
d = bench.make_data(256, 1, 1, "cuda")
with torch.no_grad():
    for tag, mutate in cases:
        p = bench.graph_net_params()
        mutate(p)
        m = bench.build_model(p, 256).cuda().eval()
        for _ in range(5):
            m(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m(d)
        torch.cuda.synchronize()
        models.append((tag, m, (time.perf_counter() - t0) / 50 * 1e3))
    for tag, m, eager in models:
        blk = GraphedForward(m).block([d] * 20, adopt_inputs=True)
        blk.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            blk.replay()
        torch.cuda.synchronize()
        print(f"{tag}: eager {eager:.4f} ms, in a HIP-graph block {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms", flush=True)

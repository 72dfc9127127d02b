import sys, time, copy, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from gnn_cca_amd import MOTMPNet
def run(tag, mutate):
    p = bench.graph_net_params()
    mutate(p)
    m = bench.build_model(p, 256).cuda().eval()
    d = bench.make_data(256, 1, 1, "cuda")
    with torch.no_grad():
        for _ in range(5): m(d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): out = m(d)
        torch.cuda.synchronize()
    print(tag, round((time.perf_counter() - t0) / 50 * 1e3, 4), "ms", flush=True)
run("mfma family (shipped)", lambda p: None)
def wide(p):
    p['encoder_feats_dict']['nodes']['resnet50']['node_out_dim'] = 64
    p['node_model_feats_dict']['fc_dims'] = [64]
def deep(p):
    p['edge_model_feats_dict']['fc_dims'] = [12, 6]
run("generic: node latent 64", wide)
run("generic: two-layer edge MLP", deep)

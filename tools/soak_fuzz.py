import sys, os, copy, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_fuzz import random_graph, Data
from conftest import GOLDEN_DIR
from oracle.mpn_oracle import NumpyOracle, load_case
from gnn_cca_amd import MOTMPNet
params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))
t0 = time.time(); total = 0; worst = 0.0
for agg, L, ncls, bf16 in [("sum", 4, 3, False), ("mean", 3, 3, False), ("max", 2, 1, False), ("sum", 4, 3, True)]:
    p = copy.deepcopy(params); p.update(node_agg_fn=agg, num_enc_steps=L, num_class_steps=ncls)
    s = dict(sd)
    if agg != "sum":
        for k in list(s):
            if k.startswith("MPNet.node_model"): s[k] = (s[k] * np.float32(4.0)).astype(np.float32)
    m = MOTMPNet(copy.deepcopy(p), None, arch); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in s.items()})
    m = m.cuda().eval(); m.edge_state_dtype = "bf16" if bf16 else "fp32"
    orc = NumpyOracle(p, arch, s, np.float32)
    rng = np.random.default_rng(hash((agg, L)) % 2**32)
    for it in range(700):
        n, ei = random_graph(rng, ["chunks", "sparse", "unsorted", "frames"][it % 4])
        x = (rng.standard_normal((n, 64)) * 0.3).astype(np.float32); ea = rng.random((ei.shape[1], 4)).astype(np.float32)
        ref = orc.forward(x, ei, ea)
        with torch.no_grad():
            out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))
        scale = max(1.0, max(float(np.abs(r).max()) for r in ref))
        for o, r in zip(out["classified_edges"], ref):
            err = float(np.abs(o.cpu().numpy() - r).max()) / scale
            worst = max(worst, err)
            assert err <= (1e-4 if bf16 else 2e-5), (agg, L, it, n, ei.shape, err)
        total += 1
    print(agg, L, bf16, "ok", total, "worst", worst, flush=True)
print("soak done", total, "graphs in", round(time.time() - t0, 1), "s; worst relative deviation", worst)

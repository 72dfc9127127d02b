#!/usr/bin/env python3
"""Timing of one training step (forward + loss + backward) on the GPU next to the autograd oracle on
the host CPU.  usage (GPU box): python3 tools/bench_train_step.py [frames] [cams] [dets_per_cam]"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnn_cca_amd import MOTMPNet  # noqa: E402
from oracle.mpn_oracle import TorchTrainOracle  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = int(sys.argv[3]) if len(sys.argv) > 3 else 5
params = bench.graph_net_params(cls_bn=False)  # config_training.yaml shape
n_g = cams * per
rows, cols = [], []
for f in range(frames):
    idx = np.arange(f * n_g, (f + 1) * n_g)
    cam = np.repeat(np.arange(cams), per)
    i, j = np.meshgrid(idx, idx, indexing="ij")
    m = cam[i - f * n_g] != cam[j - f * n_g]
    rows.append(i[m]); cols.append(j[m])
ei = np.stack([np.concatenate(rows), np.concatenate(cols)])
N, E = frames * n_g, ei.shape[1]
rng = np.random.default_rng(0)
x = rng.standard_normal((N, 2048)).astype(np.float32); x /= np.linalg.norm(x, axis=0, keepdims=True)
ea = rng.random((E, 4)).astype(np.float32)
lab = (rng.random(E) < 0.2).astype(np.float32)
model = bench.build_model(copy.deepcopy(params), n_g).cuda().train()
class D: pass
d = D(); d.x, d.edge_index, d.edge_attr = torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()
labels = torch.from_numpy(lab).cuda()
crit = torch.nn.BCEWithLogitsLoss()
opt = torch.optim.SGD(model.parameters(), lr=1e-3)
def step():
    opt.zero_grad()
    loss = sum(crit(t.view(-1), labels) for t in model(d)["classified_edges"])
    loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / 20
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
torch.set_num_threads(16)
orc = TorchTrainOracle(copy.deepcopy(params), "resnet50", sd)
orc.loss_and_grads(x, ei, ea, lab)
t0 = time.perf_counter()
for _ in range(3): orc.loss_and_grads(x, ei, ea, lab)
t_cpu = (time.perf_counter() - t0) / 3
print(json.dumps({"stage": "train step (fwd+loss+bwd+SGD)", "frames": frames, "nodes": N, "edges": E, "gpu_ms": t_gpu * 1e3,
                  "cpu_autograd_oracle_ms_16thr": t_cpu * 1e3, "speedup": t_cpu / t_gpu}))

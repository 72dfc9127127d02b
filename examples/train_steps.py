#!/usr/bin/env python3
"""Training through the HIP path (train.py:454-494 of the reference: forward, BCE over the classified steps, backward, SGD) on
synthetic frames -- once with the shipped training configuration (fused training kernels, the whole iteration replayed as one HIP
graph) and once with BatchNorm + Dropout switched on in every MLP (the layer-by-layer engine).  GPU box:
    python examples/train_steps.py
"""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (graph_net_params: the reference's GRAPH_NET_PARAMS)
from gnn_cca_amd import MOTMPNet  # noqa: E402
from gnn_cca_amd.training import GraphedTrainStep  # noqa: E402


class Batch:
    pass


def frames(n_frames, cams, per, rng):
    """Disjoint union of cross-camera graphs (inference.py:209-216) with 2048-d node features and 4 edge attributes."""
    n_g = cams * per
    rows, cols = [], []
    cam = np.repeat(np.arange(cams), per)
    for f in range(n_frames):
        i, j = np.meshgrid(np.arange(n_g), np.arange(n_g), indexing="ij")
        m = cam[i] != cam[j]
        rows.append(i[m] + f * n_g), cols.append(j[m] + f * n_g)
    ei = np.stack([np.concatenate(rows), np.concatenate(cols)])
    b = Batch()
    b.x = torch.from_numpy(rng.standard_normal((n_frames * n_g, 2048)).astype(np.float32) * 0.05).cuda()
    b.edge_index = torch.from_numpy(ei).cuda()
    b.edge_attr = torch.from_numpy(rng.random((ei.shape[1], 4)).astype(np.float32)).cuda()
    labels = torch.from_numpy((rng.random(ei.shape[1]) < 0.25).astype(np.float32)).cuda()
    return b, labels


def run(title, params, steps=30):
    torch.manual_seed(0)
    model = MOTMPNet(copy.deepcopy(params), None, "resnet50").cuda().train()
    model.set_dropout_seed(1)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    crit = torch.nn.BCEWithLogitsLoss()
    loss_fn = lambda out, lab: sum(crit(o.view(-1), lab) for o in out["classified_edges"])   # train.py:80-97
    step = GraphedTrainStep(model, opt, loss_fn, warmup=3)
    rng = np.random.default_rng(0)
    batch, labels = frames(32, 4, 6, rng)
    losses = []
    for it in range(steps):
        batch.edge_attr = torch.from_numpy(rng.random(tuple(batch.edge_attr.shape)).astype(np.float32)).cuda()
        losses.append(float(step(batch, labels)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(batch, labels)
    torch.cuda.synchronize()
    print(f"{title}: engine = {model._train_path}, loss {losses[0]:.4f} -> {losses[-1]:.4f}, {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per "
          f"training step (N = {batch.x.shape[0]}, E = {batch.edge_index.shape[1]}, replayed HIP graph)")
    assert losses[-1] < losses[0]


if __name__ == "__main__":
    run("shipped training configuration", bench.graph_net_params(cls_bn=False))
    p = bench.graph_net_params(cls_bn=True)
    p["encoder_feats_dict"]["nodes"]["resnet50"].update(use_batchnorm=True, dropout_p=0.1)
    p["edge_model_feats_dict"].update(use_batchnorm=True, dropout_p=0.1)
    p["node_model_feats_dict"].update(use_batchnorm=True, dropout_p=0.1)
    p["classifier_feats_dict"]["dropout_p"] = 0.1
    run("BatchNorm + Dropout in every MLP", p)

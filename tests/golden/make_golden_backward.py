#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f row N3 (training step through the MPN), produced by the REFERENCE's own MOTMPNet in
train mode with torch autograd.  Build container only:  python tests/golden/make_golden_backward.py

Loss as train.py:80-97 forms it: sum over the classified steps of BCEWithLogitsLoss(reduction='mean') (LOSS NAME 'BCE',
main_training.py:266-268) on synthetic edge labels.  Stored: inputs, labels, weights, logits, loss and d loss / d every
parameter.  Model shape = the shipped TRAINING config (config_training.yaml:94-181: classifier BatchNorm off), with a
64-d node input to keep the files small; one extra case keeps the classifier BatchNorm on (train-mode batch statistics).
"""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import (_Data, _install_torch_scatter_standin, build_model, cross_camera_edges, dense_edges,  # noqa: E402
                         make_inputs, make_params)


def run(MOTMPNet, name, params, arch, n, ei, seed_w, seed_in, scale):
    enc = params["encoder_feats_dict"]["nodes"][arch]
    model = build_model(MOTMPNet, params, arch, seed_w, scale)
    model.train()
    x, eit, ea = make_inputs(n, ei, enc["node_in_dim"], params["encoder_feats_dict"]["edges"]["edge_in_dim"], seed_in)
    g = torch.Generator().manual_seed(seed_in + 7)
    labels = (torch.rand(ei.shape[1], generator=g) < 0.3).float()
    data = _Data()
    data.x, data.edge_index, data.edge_attr = x, eit, ea
    sd_before = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    out = model(data)["classified_edges"]
    crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
    loss = 0
    for t in out:
        loss = loss + crit(t.view(-1), labels)
    loss.backward()
    rec = {"params_json": np.array(json.dumps({"model_params": params, "arch": arch})),
           "x": x.numpy(), "edge_index": eit.numpy(), "edge_attr": ea.numpy(), "labels": labels.numpy(),
           "loss": np.float32(loss.item()), "n_logits": np.int64(len(out))}
    for i, t in enumerate(out):
        rec[f"logits_{i}"] = t.detach().numpy()
    for k, v in sd_before.items():
        rec["sd::" + k] = v
    for k, p in model.named_parameters():
        rec["grad::" + k] = p.grad.numpy()
    for k, v in model.state_dict().items():  # BatchNorm buffers after the train-mode forward
        if "running_" in k or "num_batches" in k:
            rec["after::" + k] = v.numpy()
    np.savez(os.path.join(HERE, f"bwd_{name}.npz"), **rec)
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    print(f"bwd_{name:14s} N={n} E={ei.shape[1]} loss={loss.item():.5f} max|grad|={gmax:.4f}")


def main():
    _install_torch_scatter_standin()
    sys.path.insert(0, "/root/reference")
    from models.mpn import MOTMPNet

    tiny = dict(node_in=64, arch="tiny64", cls_bn=False)
    n, ei = cross_camera_edges([4, 4])
    run(MOTMPNet, "n8_sum", make_params(**tiny), "tiny64", n, ei, 301, 302, 0.25)
    n, ei = cross_camera_edges([8, 8, 8, 8])
    run(MOTMPNet, "terrace32", make_params(**tiny), "tiny64", n, ei, 303, 304, 1.0 / 24)
    run(MOTMPNet, "terrace32_mean", make_params(agg="mean", **tiny), "tiny64", n, ei, 305, 306, 1.0)
    run(MOTMPNet, "terrace32_max", make_params(agg="max", **tiny), "tiny64", n, ei, 311, 312, 1.0)
    run(MOTMPNet, "terrace32_reatt_n", make_params(reattach_nodes=True, **tiny), "tiny64", n, ei, 313, 314, 1.0 / 24)
    run(MOTMPNet, "terrace32_reatt_e", make_params(reattach_edges=True, **tiny), "tiny64", n, ei, 315, 316, 1.0 / 24)
    run(MOTMPNet, "terrace32_reatt_ne_mean", make_params(reattach_nodes=True, reattach_edges=True, agg="mean", **tiny), "tiny64",
        n, ei, 317, 318, 1.0)
    n, ei = dense_edges(20)
    perm = np.random.default_rng(9).permutation(ei.shape[1])
    run(MOTMPNet, "dense20_shuf", make_params(L=3, n_cls=2, **tiny), "tiny64", n, ei[:, perm], 307, 308, 1.0 / 19)
    n, ei = cross_camera_edges([5, 4, 3])
    run(MOTMPNet, "cls_bn_train", make_params(node_in=64, arch="tiny64", cls_bn=True), "tiny64", n, ei, 309, 310, 1.0 / 8)


if __name__ == "__main__":
    main()

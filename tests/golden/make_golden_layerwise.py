#!/usr/bin/env python3
"""Golden vectors for the layer-by-layer training engine (SURVEY.md 8f row N3 remainder: train-mode BatchNorm in the encoder / MPN
MLPs, Dropout, the generic family), produced by the REFERENCE's own MOTMPNet in train mode under torch autograd.  Build container
only:    python tests/golden/make_golden_layerwise.py

As in make_golden_dropout.py the reference's nn.Dropout modules are replaced by modules that apply GIVEN masks (torch's own RNG
stream cannot be reproduced by anyone else): same positions, same inverted scaling, same call order, mask values from
oracle.mpn_oracle.dropout_scale keyed by (seed, drop_stream(base of the call, layer), element).  BatchNorm1d is the reference's
own, in train mode: batch statistics, running buffers updated.  Stored: inputs, labels, weights and buffers before the step, the
probabilities and the seed, logits, loss, d loss / d every parameter, the BatchNorm buffers after the step.
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import (_Data, _install_torch_scatter_standin, build_model, cross_camera_edges, dense_edges,  # noqa: E402
                         make_inputs, make_params)
from oracle.mpn_oracle import (DROP_CLS, DROP_EDGE_STEP, DROP_ENC_EDGE, DROP_ENC_NODE1, DROP_NODE_STEP, drop_stream,  # noqa: E402
                               dropout_scale)


class InjectedDropout(torch.nn.Module):
    """Stands where an nn.Dropout stood in the reference's Sequential: layer `li` of an MLP whose k-th call has the base stream
    first_base + k * per_call."""

    def __init__(self, p, seed, first_base, per_call, li):
        super().__init__()
        self.p, self.seed, self.first, self.per_call, self.li, self.calls = p, seed, first_base, per_call, li, 0

    def forward(self, x):
        stream = drop_stream(self.first + self.calls * self.per_call, self.li)
        self.calls += 1
        if self.p <= 0 or not self.training:
            return x
        return x * torch.from_numpy(dropout_scale(self.seed, stream, x.shape[0], x.shape[1], self.p))


def inject(model, seed):
    def swap(mlp, first_base, per_call):
        if mlp is None:
            return
        li = -1
        for i, mod in enumerate(mlp.fc_layers):
            if isinstance(mod, torch.nn.Linear):
                li += 1
            if isinstance(mod, torch.nn.Dropout):
                mlp.fc_layers[i] = InjectedDropout(mod.p, seed, first_base, per_call, li)
    swap(model.encoder.node_mlp, DROP_ENC_NODE1, 0)
    swap(model.encoder.edge_mlp, DROP_ENC_EDGE, 0)
    swap(model.MPNet.edge_model.edge_mlp, DROP_EDGE_STEP + 1, 1)
    swap(model.MPNet.node_model.node_mlp, DROP_NODE_STEP + 1, 1)
    swap(model.classifier.edge_mlp, DROP_CLS, 1)


def run(MOTMPNet, name, params, arch, n, ei, seed_w, seed_in, scale, ps, seed):
    enc = params["encoder_feats_dict"]["nodes"][arch]
    enc["dropout_p"] = ps[0]
    params["edge_model_feats_dict"]["dropout_p"] = ps[1]
    params["node_model_feats_dict"]["dropout_p"] = ps[2]
    params["classifier_feats_dict"]["dropout_p"] = ps[3]
    model = build_model(MOTMPNet, params, arch, seed_w, scale)
    inject(model, seed)
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
           "loss": np.float32(loss.item()), "n_logits": np.int64(len(out)), "dropout_p": np.asarray(ps, dtype=np.float32),
           "dropout_seed": np.int64(seed)}
    for i, t in enumerate(out):
        rec[f"logits_{i}"] = t.detach().numpy()
    for k, v in sd_before.items():
        rec["sd::" + k] = v
    for k, p in model.named_parameters():
        rec["grad::" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            rec["after::" + k] = v.numpy()
    np.savez(os.path.join(HERE, f"lw_{name}.npz"), **rec)
    gmax = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
    print(f"lw_{name:24s} N={n} E={ei.shape[1]} loss={loss.item():.5f} max|grad|={gmax:.4f} p={ps}")


def main():
    _install_torch_scatter_standin()
    sys.path.insert(0, "/root/reference")
    from models.mpn import MOTMPNet

    # shipped widths, BatchNorm switched on in the encoder and in both MPN MLPs (config keys use_batchnorm), classifier BN on
    n, ei = cross_camera_edges([8, 8, 8, 8])
    bn = dict(node_in=64, arch="tiny64", cls_bn=True, enc_bn=True, mpn_bn=True)
    run(MOTMPNet, "bn_everywhere", make_params(**bn), "tiny64", n, ei, 503, 504, 1.0, (0.0, 0.0, 0.0, 0.0), 7001)
    run(MOTMPNet, "bn_drop_mean", make_params(agg="mean", **bn), "tiny64", n, ei, 505, 506, 1.0, (0.2, 0.1, 0.3, 0.25), 7002)
    run(MOTMPNet, "bn_mpn_only_max", make_params(node_in=64, arch="tiny64", cls_bn=False, mpn_bn=True, agg="max"), "tiny64", n, ei,
        507, 508, 1.0, (0.0, 0.0, 0.2, 0.0), 7003)
    # the generic family: other widths, multi-layer MLPs, BatchNorm + Dropout in every MLP
    n, ei = cross_camera_edges([5, 4, 4])
    run(MOTMPNet, "generic_dims",
        make_params(node_in=40, node_fc=(48, 24), node_out=16, edge_in=3, edge_fc=(5,), edge_out=8, edge_mlp_fc=(10, 8),
                    node_mlp_fc=(24, 16), cls_fc=(5, 3), cls_bn=True, enc_bn=True, mpn_bn=True, arch="generic"),
        "generic", n, ei, 511, 512, 1.0, (0.3, 0.2, 0.25, 0.15), 7004)
    # both reattach flags, 'max', single-layer encoder, bare Linear classifier, unsorted ragged graph (isolated nodes, duplicates)
    rng = np.random.default_rng(17)
    n = 18
    ei = np.stack([rng.integers(0, 15, size=70), rng.integers(0, 18, size=70)]).astype(np.int64)
    run(MOTMPNet, "generic_reattach_max",
        make_params(node_in=24, node_fc=(), node_out=12, edge_in=3, edge_fc=(), edge_out=5, edge_mlp_fc=(7, 5), node_mlp_fc=(12,),
                    cls_fc=(), cls_bn=False, agg="max", reattach_nodes=True, reattach_edges=True, arch="generic"),
        "generic", n, ei, 513, 514, 1.0, (0.0, 0.1, 0.2, 0.0), 7005)
    # shuffled dense graph, generic widths without BatchNorm, L = 3 with 2 classified steps, 'sum'
    n, ei = dense_edges(12)
    perm = np.random.default_rng(9).permutation(ei.shape[1])
    run(MOTMPNet, "generic_dense12_shuf",
        make_params(node_in=32, node_fc=(20,), node_out=10, edge_in=4, edge_fc=(), edge_out=4, edge_mlp_fc=(4,), node_mlp_fc=(10,),
                    cls_fc=(3,), cls_bn=False, L=3, n_cls=2, arch="generic"),
        "generic", n, ei[:, perm], 515, 516, 1.0 / 11, (0.1, 0.0, 0.0, 0.2), 7006)
    # L = 0: the classifier on the encoded edge features only (models/mpn.py:295-297)
    n, ei = cross_camera_edges([4, 4])
    run(MOTMPNet, "L0_bn", make_params(node_in=64, arch="tiny64", cls_bn=True, L=0, n_cls=0), "tiny64", n, ei, 517, 518, 1.0,
        (0.1, 0.0, 0.0, 0.2), 7007)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for train-mode Dropout (SURVEY.md 8f row N3 remainder), produced by the REFERENCE's own MOTMPNet in train mode
under torch autograd, with its nn.Dropout modules replaced by modules that apply GIVEN masks.  Build container only:
    python tests/golden/make_golden_dropout.py

torch's own Dropout draws its masks from a RNG stream no other implementation can reproduce, so "the reference with Dropout"
is pinned the only way it can be: same positions (behind the ReLU of every MLP layer wider than 1, models/mlp.py:17-21), same
inverted scaling 1 / (1 - p), same call order (one call of edge_mlp / node_mlp per step, one of the classifier per classified
step) -- with the mask VALUES injected.  The masks come from oracle.mpn_oracle.dropout_scale, the numpy twin of the HIP kernels'
counter-based generator, keyed by (seed, tensor id, element).  Stored: inputs, labels, weights, the four probabilities and the
seed, logits, loss, d loss / d every parameter.
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
from oracle.mpn_oracle import (DROP_CLS, DROP_EDGE_STEP, DROP_ENC_EDGE, DROP_ENC_NODE1, DROP_ENC_NODE2, DROP_NODE_STEP,  # noqa: E402
                               dropout_scale)


class InjectedDropout(torch.nn.Module):
    """Stands where an nn.Dropout stood in the reference's Sequential: multiplies by the mask of (seed, stream) -- the stream of
    its k-th call is first_stream + k * per_call."""

    def __init__(self, p, seed, first_stream, per_call):
        super().__init__()
        self.p, self.seed, self.first, self.per_call, self.calls = p, seed, first_stream, per_call, 0

    def forward(self, x):
        stream = self.first + self.calls * self.per_call
        self.calls += 1
        if self.p <= 0 or not self.training:
            return x
        return x * torch.from_numpy(dropout_scale(self.seed, stream, x.shape[0], x.shape[1], self.p))


def inject(model, seed):
    def swap(mlp, streams, per_call):
        it = iter(streams)
        for i, mod in enumerate(mlp.fc_layers):
            if isinstance(mod, torch.nn.Dropout):
                mlp.fc_layers[i] = InjectedDropout(mod.p, seed, next(it), per_call)
    swap(model.encoder.node_mlp, [DROP_ENC_NODE1, DROP_ENC_NODE2], 0)
    swap(model.encoder.edge_mlp, [DROP_ENC_EDGE], 0)
    swap(model.MPNet.edge_model.edge_mlp, [DROP_EDGE_STEP + 1], 1)      # step 1, 2, ...
    swap(model.MPNet.node_model.node_mlp, [DROP_NODE_STEP + 1], 1)
    swap(model.classifier.edge_mlp, [DROP_CLS], 1)                      # classified step 0, 1, ...


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
        rec["grad::" + k] = p.grad.numpy()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            rec["after::" + k] = v.numpy()
    np.savez(os.path.join(HERE, f"drop_{name}.npz"), **rec)
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    print(f"drop_{name:22s} N={n} E={ei.shape[1]} loss={loss.item():.5f} max|grad|={gmax:.4f} p={ps}")


def main():
    _install_torch_scatter_standin()
    sys.path.insert(0, "/root/reference")
    from models.mpn import MOTMPNet

    tiny = dict(node_in=64, arch="tiny64", cls_bn=False)
    n, ei = cross_camera_edges([8, 8, 8, 8])
    run(MOTMPNet, "terrace32", make_params(**tiny), "tiny64", n, ei, 403, 404, 1.0 / 24, (0.2, 0.1, 0.3, 0.25), 9001)
    run(MOTMPNet, "terrace32_max", make_params(agg="max", **tiny), "tiny64", n, ei, 411, 412, 1.0, (0.1, 0.1, 0.4, 0.1), 9002)
    run(MOTMPNet, "terrace32_reatt_mean", make_params(reattach_nodes=True, reattach_edges=True, agg="mean", **tiny), "tiny64",
        n, ei, 417, 418, 1.0, (0.2, 0.2, 0.2, 0.2), 9003)
    n, ei = cross_camera_edges([5, 4, 3])
    run(MOTMPNet, "cls_bn", make_params(node_in=64, arch="tiny64", cls_bn=True), "tiny64", n, ei, 409, 410, 1.0 / 8,
        (0.15, 0.1, 0.2, 0.3), 9004)
    n, ei = dense_edges(20)
    perm = np.random.default_rng(9).permutation(ei.shape[1])
    run(MOTMPNet, "dense20_shuf", make_params(L=3, n_cls=2, **tiny), "tiny64", n, ei[:, perm], 407, 408, 1.0 / 19,
        (0.2, 0.1, 0.3, 0.2), 9005)


if __name__ == "__main__":
    main()

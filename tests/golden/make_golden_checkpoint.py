#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f row N4 (checkpoint loading), produced by the reference's own
`libs.utils.load_pretrained_weights` (libs/utils.py:458-507) applied to the reference's own `MOTMPNet`.
Build container only:  python tests/golden/make_golden_checkpoint.py

`libs/utils.py` imports `cv2` and `torch_scatter`, absent here; neither is touched by the two functions used, so empty
stand-in modules are registered before the import.  The synthetic checkpoint (written to a temp dir) is what
`utils.save_checkpoint` would store: {'model_state_dict': ..., 'epoch': ...}, with DataParallel's 'module.' prefix on
the keys, one tensor of the wrong size and one unknown key.  Stored: the checkpoint tensors, the state_dict of the
reference model after the reference loaded them, and (round 3) a small cross-camera graph with the logits of that loaded
reference model in eval mode -- what a checkpoint converted to a packed blob must reproduce on the GPU.
`ckpt_bn_variant.npz`: a checkpoint written by a model WITH the classifier BatchNorm (config_inference.yaml:163) loaded into a
model WITHOUT it (config_training.yaml:181): the classifier's second Linear has another Sequential index there, so the
reference's loader drops it (and the BatchNorm tensors) and keeps the initial values.
"""
import copy
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _Data, _install_torch_scatter_standin, cross_camera_edges, make_inputs, make_params  # noqa: E402


def _forward_record(model, rec, seed):
    """Inputs of a 3-camera frame (4 detections each, E = 96) and the loaded reference model's eval-mode logits."""
    n, ei = cross_camera_edges([4, 4, 4])
    x, ei, ea = make_inputs(n, ei, 64, 4, seed)
    d = _Data()
    d.x, d.edge_index, d.edge_attr = x, ei, ea
    model.eval()
    with torch.no_grad():
        out = model(d)["classified_edges"]
    rec["x"], rec["edge_index"], rec["edge_attr"] = x.numpy(), ei.numpy(), ea.numpy()
    for i, o in enumerate(out):
        rec[f"logits_{i}"] = o.numpy()
    return max(float(o.abs().max()) for o in out)


def main():
    _install_torch_scatter_standin()
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, "/root/reference")
    from libs import utils  # the reference, unmodified
    from models.mpn import MOTMPNet

    params = make_params(node_in=64, arch="tiny64")
    torch.manual_seed(7)
    src = MOTMPNet(copy.deepcopy(params), None, "tiny64")  # the "trained" model
    with torch.no_grad():
        for p in src.MPNet.node_model.node_mlp.parameters():
            p.mul_(1.0 / 8)                                    # conditioned as the other goldens
        g = torch.Generator().manual_seed(70)
        for mod in src.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):         # a trained BatchNorm has non-trivial statistics
                mod.running_mean.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.weight.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.num_batches_tracked.fill_(321)
    ckpt_sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    ckpt_sd["module.encoder.node_mlp.fc_layers.3.bias"] = torch.randn(7)       # wrong size -> discarded
    ckpt_sd["module.some.unknown.tensor"] = torch.randn(3)                     # unknown name -> discarded
    ckpt = {"epoch": 12, "model_state_dict": ckpt_sd, "prec": 91.0}
    torch.manual_seed(8)
    dst = MOTMPNet(copy.deepcopy(params), None, "tiny64")  # differently initialised target
    init_sd = {k: v.clone() for k, v in dst.state_dict().items()}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt_latest.pth.tar")
        torch.save(ckpt, path)
        utils.load_pretrained_weights(dst, path)
    rec = {}
    for k, v in ckpt_sd.items():
        rec["ckpt::" + k] = v.numpy()
    for k, v in init_sd.items():
        rec["init::" + k] = v.numpy()
    for k, v in dst.state_dict().items():
        rec["loaded::" + k] = v.numpy()
    mx = _forward_record(dst, rec, 71)
    np.savez(os.path.join(HERE, "ckpt_module_prefix.npz"), **rec)
    print("ckpt_module_prefix: %d checkpoint tensors, %d model tensors, max|logit| %.3f" % (len(ckpt_sd), len(init_sd), mx))

    # --- BatchNorm-on checkpoint into a BatchNorm-off classifier ---------------------------------------------------------
    params_off = make_params(node_in=64, arch="tiny64", cls_bn=False)
    torch.manual_seed(9)
    dst2 = MOTMPNet(copy.deepcopy(params_off), None, "tiny64")
    with torch.no_grad():
        for p in dst2.MPNet.node_model.node_mlp.parameters():
            p.mul_(1.0 / 8)
    init2 = {k: v.clone() for k, v in dst2.state_dict().items()}
    ckpt2_sd = {k: v.clone() for k, v in src.state_dict().items()}          # no 'module.' prefix this time
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt_best.pth.tar")
        torch.save({"epoch": 3, "model_state_dict": ckpt2_sd}, path)
        utils.load_pretrained_weights(dst2, path)
    rec2 = {}
    for k, v in ckpt2_sd.items():
        rec2["ckpt::" + k] = v.numpy()
    for k, v in init2.items():
        rec2["init::" + k] = v.numpy()
    for k, v in dst2.state_dict().items():
        rec2["loaded::" + k] = v.numpy()
    mx = _forward_record(dst2, rec2, 72)
    np.savez(os.path.join(HERE, "ckpt_bn_variant.npz"), **rec2)
    print("ckpt_bn_variant: %d checkpoint tensors, %d model tensors, max|logit| %.3f" % (len(ckpt2_sd), len(init2), mx))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f row N4 (checkpoint loading), produced by the reference's own
`libs.utils.load_pretrained_weights` (libs/utils.py:458-507) applied to the reference's own `MOTMPNet`.
Build container only:  python tests/golden/make_golden_checkpoint.py

`libs/utils.py` imports `cv2` and `torch_scatter`, absent here; neither is touched by the two functions used, so empty
stand-in modules are registered before the import.  The synthetic checkpoint (written to a temp dir) is what
`utils.save_checkpoint` would store: {'model_state_dict': ..., 'epoch': ...}, with DataParallel's 'module.' prefix on
the keys, one tensor of the wrong size and one unknown key.  Stored: the checkpoint tensors, and the state_dict of the
reference model after the reference loaded them.
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
from make_golden import _install_torch_scatter_standin, make_params  # noqa: E402


def main():
    _install_torch_scatter_standin()
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, "/root/reference")
    from libs import utils  # the reference, unmodified
    from models.mpn import MOTMPNet

    params = make_params(node_in=64, arch="tiny64")
    torch.manual_seed(7)
    src = MOTMPNet(copy.deepcopy(params), None, "tiny64")  # the "trained" model
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
    np.savez(os.path.join(HERE, "ckpt_module_prefix.npz"), **rec)
    print("ckpt_module_prefix: %d checkpoint tensors, %d model tensors" % (len(ckpt_sd), len(init_sd)))


if __name__ == "__main__":
    main()

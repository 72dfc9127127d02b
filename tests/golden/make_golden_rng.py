#!/usr/bin/env python3
"""Golden for the RNG contract of construction: what torch's GLOBAL generator yields right after `MOTMPNet(...)` was built under a
seed.  The reference's `_build_core_MPNet` also builds an unused `node_mlp_old` (models/mpn.py:241-242) -- it registers nothing but
draws from the generator after every real parameter exists, so a seeded training script sees a different stream afterwards unless a
replacement draws the same amount.  Stored per case: the seed, eight `torch.rand` values drawn right after construction, and a
checksum of every Linear parameter (the parameters themselves are pinned by the case's own golden).

Run in the build container only (imports the reference from /root/reference, like make_golden.py):
    python tests/golden/make_golden_rng.py
"""
import copy
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (the torch_scatter stand-in and the params builder)


def main():
    mg._install_torch_scatter_standin()
    sys.path.insert(0, mg.REF)
    from models.mpn import MOTMPNet  # the reference itself
    cases = {
        "default_bn": (mg.make_params(), "resnet50", 11),
        "bn_off_mean": (mg.make_params(cls_bn=False, agg="mean"), "resnet50", 3),
        "reattach": (mg.make_params(reattach_nodes=True, reattach_edges=True, L=2, n_cls=2), "resnet50", 5),
        "generic": (mg.make_params(node_in=96, node_fc=(64, 48), node_out=40, edge_out=10, edge_mlp_fc=(12, 10), node_mlp_fc=(40,),
                                   cls_fc=(8, 4), enc_bn=True), "resnet50", 7),
    }
    out = {}
    for name, (params, arch, seed) in cases.items():
        torch.manual_seed(seed)
        model = MOTMPNet(copy.deepcopy(params), None, arch)
        after = torch.rand(8)
        h = hashlib.sha256()
        for k, v in model.state_dict().items():
            h.update(k.encode())
            h.update(v.detach().cpu().contiguous().numpy().tobytes())
        out[f"{name}::seed"] = np.int64(seed)
        out[f"{name}::after"] = after.numpy()
        out[f"{name}::state_sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)
        out[f"{name}::params_json"] = np.array(mg.json_dumps(params) if hasattr(mg, "json_dumps") else __import__("json").dumps(params))
        out[f"{name}::arch"] = np.array(arch)
    np.savez(os.path.join(HERE, "rng_after_init.npz"), **out)
    print("wrote rng_after_init.npz:", sorted(k for k in out if k.endswith("::after")))


if __name__ == "__main__":
    main()

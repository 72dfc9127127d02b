"""Row N4: gnn_cca_amd.checkpoint.load_pretrained_weights against the state the REFERENCE's own loader leaves its own
model in (tests/golden/ckpt_module_prefix.npz).  CPU only."""
import copy
import os

import numpy as np
import torch

from conftest import GOLDEN_DIR


def _params():
    import json
    z = np.load(os.path.join(GOLDEN_DIR, "n8_sum.npz"), allow_pickle=False)
    meta = json.loads(str(z["params_json"]))
    return meta["model_params"], meta["arch"]


def test_loader_matches_reference_loader(tmp_path):
    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.checkpoint import checkpoint_to_blob, load_pretrained_weights
    z = np.load(os.path.join(GOLDEN_DIR, "ckpt_module_prefix.npz"))
    ckpt = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("ckpt::")}
    init = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init::")}
    want = {k[8:]: z[k] for k in z.files if k.startswith("loaded::")}
    params, arch = _params()
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict(init, strict=True)               # same starting point as the reference's target model
    path = os.path.join(tmp_path, "ckpt_latest.pth.tar")
    torch.save({"epoch": 12, "model_state_dict": ckpt}, path)
    m, rep = load_pretrained_weights(m, path, verbose=False)
    got = m.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert np.array_equal(got[k].numpy(), want[k]), k
    assert sorted(rep.discarded) == ["encoder.node_mlp.fc_layers.3.bias", "some.unknown.tensor"]
    assert rep.missing == ["encoder.node_mlp.fc_layers.3.bias"]
    # the mismatched tensor keeps its initial value (reference behaviour)
    assert np.array_equal(got["encoder.node_mlp.fc_layers.3.bias"].numpy(), init["encoder.node_mlp.fc_layers.3.bias"].numpy())
    # checkpoint -> blob equals packing the loaded module
    blob, _ = checkpoint_to_blob({"model_state_dict": {k: v for k, v in got.items()}}, copy.deepcopy(params), arch)
    assert torch.equal(blob, m.eval().pack_weights_host())


def test_raw_state_dict_and_nothing_matching():
    import warnings

    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.checkpoint import load_pretrained_weights
    params, arch = _params()
    a = MOTMPNet(copy.deepcopy(params), None, arch)
    b = MOTMPNet(copy.deepcopy(params), None, arch)
    b, rep = load_pretrained_weights(b, a.state_dict(), verbose=False)   # a bare state_dict is accepted too
    assert not rep.discarded and not rep.missing
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, rep = load_pretrained_weights(b, {"nope": torch.zeros(1)}, verbose=False)
    assert not rep.matched and len(w) == 1

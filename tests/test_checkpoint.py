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


def test_packed_blob_header_is_validated_on_load():
    """A blob persisted by the converter is only valid for the build and configuration that packed it."""
    import copy

    import numpy as np
    import pytest
    import torch

    from conftest import GOLDEN_DIR
    from gnn_cca_amd import MOTMPNet
    from oracle.mpn_oracle import load_case
    import os
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    good = m.pack_weights_host()
    m.load_packed_blob(good.numpy().tobytes())          # accepted (CPU module: installs a host copy)
    assert m._packed is not None and torch.equal(m._packed[1].cpu(), good)
    stale = good.clone()
    stale[0] ^= 0x01                                     # another layout generation
    with pytest.raises(RuntimeError, match="different build"):
        m.load_packed_blob(stale)
    with pytest.raises(RuntimeError, match="bytes"):
        m.load_packed_blob(good[:-4])
    p2 = copy.deepcopy(params)
    p2["num_enc_steps"] = 2
    other, _, _, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))
    m2 = MOTMPNet(copy.deepcopy(other), None, "tiny64")
    with pytest.raises(RuntimeError):
        m2.load_packed_blob(good)                        # a different GRAPH_NET_PARAMS: different size


def test_plan_load_is_pure_and_explains():
    from gnn_cca_amd.checkpoint import plan_load
    shapes = {"a.weight": (2, 3), "a.bias": (2,), "b": ()}
    ck = {"module.a.weight": torch.zeros(2, 3), "a.bias": torch.zeros(3), "zzz": torch.zeros(1), "module.module.b": torch.zeros(())}
    accepted, rep = plan_load(shapes, ck)
    assert list(accepted) == ["a.weight"] and rep.matched == ["a.weight"]
    assert rep.discarded == ["a.bias", "zzz", "module.b"]      # only ONE DataParallel prefix is removed
    assert "shape (3,)" in rep.reasons["a.bias"] and rep.reasons["zzz"] == "not a tensor of this model"
    assert rep.missing == ["a.bias", "b"]
    assert "skipped a.bias" in rep.summary()

"""Row N3: the autograd oracle (oracle.TorchTrainOracle) against gradients produced by the reference's own module
under torch autograd (tests/golden/bwd_*.npz).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle.mpn_oracle import TorchTrainOracle

CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "bwd_*.npz")))


DROP_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "drop_*.npz")))
LW_CASES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "lw_*.npz")))


def load_bwd(name, prefix="bwd_"):
    z = np.load(os.path.join(GOLDEN_DIR, f"{prefix}{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["params_json"]))
    sd = {k[4:]: z[k] for k in z.files if k.startswith("sd::")}
    grads = {k[6:]: z[k] for k in z.files if k.startswith("grad::")}
    after = {k[7:]: z[k] for k in z.files if k.startswith("after::")}
    return meta["model_params"], meta["arch"], sd, grads, after, {k: z[k] for k in z.files if "::" not in k}


@pytest.mark.parametrize("name", CASES)
def test_autograd_oracle_matches_reference(name):
    params, arch, sd, grads, after, a = load_bwd(name)
    orc = TorchTrainOracle(params, arch, sd)
    loss, logits, g = orc.loss_and_grads(a["x"], a["edge_index"], a["edge_attr"], a["labels"])
    assert abs(loss - float(a["loss"])) <= 2e-6
    for i, t in enumerate(logits):
        assert np.abs(t.numpy() - a[f"logits_{i}"]).max() <= 2e-6
    assert sorted(g) == sorted(grads)
    for k in grads:
        scale = max(1.0, float(np.abs(grads[k]).max()))
        assert np.abs(g[k].numpy() - grads[k]).max() <= 2e-6 * scale, k
    for k, v in after.items():  # BatchNorm running statistics after the train-mode forward
        if "running_" in k:
            assert np.abs(orc.buffers[k].numpy() - v).max() <= 1e-6, k


@pytest.mark.parametrize("name", DROP_CASES)
def test_autograd_oracle_with_dropout_matches_reference(name):
    """tests/golden/drop_*.npz: the reference's module in train mode with its nn.Dropout modules applying INJECTED masks
    (tests/golden/make_golden_dropout.py).  The oracle, given the same probabilities and seed, must place the same masks at the
    same layers with the same scaling and reproduce logits, loss and every gradient."""
    params, arch, sd, grads, after, a = load_bwd(name, "drop_")
    ps = [float(v) for v in a["dropout_p"]]
    orc = TorchTrainOracle(params, arch, sd, dropout=dict(p_enc=ps[0], p_edge=ps[1], p_node=ps[2], p_cls=ps[3],
                                                          seed=int(a["dropout_seed"])))
    loss, logits, g = orc.loss_and_grads(a["x"], a["edge_index"], a["edge_attr"], a["labels"])
    assert abs(loss - float(a["loss"])) <= 2e-6
    for i, t in enumerate(logits):
        assert np.abs(t.numpy() - a[f"logits_{i}"]).max() <= 2e-6
    for k in grads:
        scale = max(1.0, float(np.abs(grads[k]).max()))
        assert np.abs(g[k].numpy() - grads[k]).max() <= 2e-6 * scale, k


@pytest.mark.parametrize("name", LW_CASES)
def test_autograd_oracle_matches_reference_bn_dropout_generic(name):
    """tests/golden/lw_*.npz (make_golden_layerwise.py): the reference's module in train mode with BatchNorm in the encoder / MPN
    MLPs, injected Dropout masks in multi-layer MLPs (stream of layer li = drop_stream(base, li)) and the generic family's widths.
    Pins the oracle the layer-by-layer HIP engine is checked against: logits, loss, every gradient, the BatchNorm buffers."""
    params, arch, sd, grads, after, a = load_bwd(name, "lw_")
    ps = [float(v) for v in a["dropout_p"]]
    orc = TorchTrainOracle(params, arch, sd, dropout=dict(p_enc=ps[0], p_edge=ps[1], p_node=ps[2], p_cls=ps[3],
                                                          seed=int(a["dropout_seed"])))
    loss, logits, g = orc.loss_and_grads(a["x"], a["edge_index"], a["edge_attr"], a["labels"])
    assert abs(loss - float(a["loss"])) <= 2e-6
    for i, t in enumerate(logits):
        assert np.abs(t.numpy() - a[f"logits_{i}"]).max() <= 2e-6
    for k in grads:
        scale = max(1.0, float(np.abs(grads[k]).max()))
        assert np.abs(g[k].numpy() - grads[k]).max() <= 2e-6 * scale, k
    for k, v in after.items():
        if "running_" in k:
            assert np.abs(orc.buffers[k].numpy() - v).max() <= 1e-6, k

"""Pins the CPU oracle (oracle/mpn_oracle.py) to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_cases
from oracle.mpn_oracle import NumpyOracle, TorchOracle, load_case, model_layout

TOL32 = 2e-6  # fp32 restatement vs fp32 reference: same ops, BLAS summation order may differ
TOL64 = 5e-6  # fp64 restatement vs fp32 reference: the reference's own rounding error


@pytest.mark.parametrize("name", golden_cases())
def test_numpy_oracle_matches_reference(name):
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    trace = {}
    out = NumpyOracle(params, arch, sd, np.float32).forward(a["x"], a["edge_index"], a["edge_attr"], trace)
    assert len(out) == int(a["n_logits"])
    for i, o in enumerate(out):
        assert o.shape == a[f"logits_{i}"].shape and o.dtype == np.float32
        assert np.abs(o - a[f"logits_{i}"]).max() <= TOL32, (name, i)
    for k, v in trace.items():  # encoder outputs and per-step latents
        assert np.abs(v - a[k]).max() <= TOL32, (name, k)


@pytest.mark.parametrize("name", golden_cases())
def test_torch_oracle_matches_reference(name):
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = TorchOracle(params, arch, sd).forward(a["x"], a["edge_index"], a["edge_attr"])
    assert len(out) == int(a["n_logits"])
    for i, o in enumerate(out):
        assert np.abs(o.numpy() - a[f"logits_{i}"]).max() <= TOL32, (name, i)


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_matches_reference(name):
    """oracle/mpn_oracle_c.c (the fused split-weight C / OpenMP restatement, SURVEY.md 8d's CPU-baseline flavour (ii)) against
    the reference's golden logits; the two goldens with multi-layer MPN MLPs are outside its scope and must say so."""
    from oracle.c_oracle import COracle
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    orc = COracle(params, arch, sd)
    if not orc.supported():
        assert name in ("generic_dims", "generic_reattach_max")
        with pytest.raises(RuntimeError):
            orc.forward(a["x"], a["edge_index"], a["edge_attr"])
        return
    out = orc.forward(a["x"], a["edge_index"], a["edge_attr"])
    assert len(out) == int(a["n_logits"])
    for i, o in enumerate(out):
        assert o.shape == a[f"logits_{i}"].shape and o.dtype == np.float32
        assert np.abs(o - a[f"logits_{i}"]).max() <= TOL32, (name, i)


@pytest.mark.parametrize("name", ["dense64", "terrace32", "ragged_max", "generic_dims"])
def test_fp64_oracle_brackets_reference(name):
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = NumpyOracle(params, arch, sd, np.float64).forward(a["x"], a["edge_index"], a["edge_attr"])
    for i, o in enumerate(out):
        assert np.abs(o - a[f"logits_{i}"]).max() <= TOL64, (name, i)


def test_layout_matches_reference_state_dict_keys():
    """The Sequential numbering rule of models/mlp.py fixes the state_dict keys."""
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    lay = model_layout(params, arch)
    assert [(l[0], l[1], l[2], l[3]) for l in lay["encoder.node_mlp"]] == [(0, 2048, 128, None), (3, 128, 32, None)]
    assert [(l[0], l[1], l[2], l[3]) for l in lay["classifier.edge_mlp"]] == [(0, 6, 4, 1), (4, 4, 1, None)]
    assert lay["MPNet.edge_model.edge_mlp"][0][1:3] == (70, 6)
    assert lay["MPNet.node_model.node_mlp"][0][1:3] == (38, 32)
    for prefix in ("encoder.node_mlp", "encoder.edge_mlp", "classifier.edge_mlp",
                   "MPNet.edge_model.edge_mlp", "MPNet.node_model.node_mlp"):
        for lin, i, o, bn, _ in lay[prefix]:
            assert sd[f"{prefix}.fc_layers.{lin}.weight"].shape == (o, i)
            if bn is not None:
                assert f"{prefix}.fc_layers.{bn}.running_var" in sd


def test_shuffled_edges_permute_logits():
    """Permutation-of-edges equivariance, on the reference's own outputs (property used by the GPU tests)."""
    _, _, _, a = load_case(os.path.join(GOLDEN_DIR, "dense24_sorted.npz"))
    _, _, _, b = load_case(os.path.join(GOLDEN_DIR, "dense24_shuffled.npz"))
    key = lambda ei: ei[0] * 1000 + ei[1]
    order_a, order_b = np.argsort(key(a["edge_index"])), np.argsort(key(b["edge_index"]))
    for i in range(3):
        assert np.abs(a[f"logits_{i}"][order_a] - b[f"logits_{i}"][order_b]).max() <= 1e-6

"""Parity at the dynamic range an UNconditioned `sum` model has (SURVEY.md 7.3 / 8(d) "report relative error too").

Every other golden / fuzz / bench parity figure uses node-MLP weights conditioned by 1/(N - 1), i.e. max |logit| ~ 0.5.  With torch's default
initialisation left as it is, `node_agg_fn: sum` grows the activations by ~N per step: SURVEY 7.3 measured max |logit| = 3.8e1 at N = 32 and
4.1e2 at N = 64 (this file's draw: 1.0e1 and 7.4e1 -- 20 to 150 times the conditioned models' 0.5), where "1e-4 absolute" stops meaning anything and the reference's OWN fp32 arithmetic is 1e-6..1e-5 relative away from fp64.
The yardstick here is therefore the fp64 oracle, the error is stated RELATIVE to max |logit|, and the bound is the reference arithmetic's own
gap: the HIP path (fp32 state) may be at most 4x as far from fp64 as the fp32 oracle (op-for-op the reference) is.  bf16 edge-state storage
rounds every stored latent to 8 significant bits: its measured relative error is stated and bounded as a number of its own.
Covers the f32-MFMA message form (single graphs), the split-bf16 message form and the fp16-split encoder GEMM (the 64-graph union: 4096 nodes)."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle.mpn_oracle import NumpyOracle, load_case

pytestmark = pytest.mark.gpu

BF16_STATE_REL_BOUND = 2e-4     # measured on MI355X (round 6, profiles/r06_logs/t1_dynamic_range.log): 2.8e-5 .. 4.7e-5 relative to max |logit|
                                # (fp32 state: 6.6e-7 .. 1.4e-6 against the fp32 oracle's own 6.3e-7 .. 1.9e-6)


class Data:
    def __init__(self, x, edge_index, edge_attr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr


def _dense(n, graphs=1):
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    k = i != j
    one = np.stack([i[k], j[k]]).astype(np.int64)
    return np.concatenate([one + g * n for g in range(graphs)], axis=1)


def _unconditioned():
    """torch's default Linear init under a seed, exactly as the golden generator drew it -- dense64.npz stores the node-MLP tensors scaled by
    1/63; the scaling is undone here, nothing else is touched."""
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    sd = dict(sd)
    for k in list(sd):
        if k.startswith("MPNet.node_model"):
            sd[k] = (np.asarray(sd[k], dtype=np.float64) * 63.0).astype(np.float32)
    return copy.deepcopy(params), arch, sd


def _run(n, graphs, state):
    from gnn_cca_amd import MOTMPNet
    params, arch, sd = _unconditioned()
    rng = np.random.default_rng(1000 * n + graphs)
    x = rng.standard_normal((n * graphs, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)          # inference.py:189-190
    ei = _dense(n, graphs)
    ea = rng.random((ei.shape[1], 4)).astype(np.float32)
    ref64 = NumpyOracle(params, arch, sd, np.float64).forward(x, ei, ea)
    ref32 = NumpyOracle(params, arch, sd, np.float32).forward(x, ei, ea)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.edge_state_dtype = state
    with torch.no_grad():
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))["classified_edges"]
    torch.cuda.synchronize()
    scale = max(float(np.abs(r).max()) for r in ref64)
    hip = max(float(np.abs(o.cpu().numpy().astype(np.float64) - r).max()) for o, r in zip(out, ref64)) / scale
    gap32 = max(float(np.abs(a.astype(np.float64) - r).max()) for a, r in zip(ref32, ref64)) / scale
    assert all(torch.isfinite(o).all().item() for o in out)
    return scale, hip, gap32


@pytest.mark.parametrize("n,graphs", [(32, 1), (64, 1), (64, 64)])
def test_unconditioned_sum_model_fp32_state_vs_fp64(n, graphs):
    scale, hip, gap32 = _run(n, graphs, "fp32")
    print(f"\n[dynamic range] N={n} x{graphs} fp32 state: max|logit| {scale:.3e}, HIP vs fp64 {hip:.3e} rel, fp32 oracle vs fp64 {gap32:.3e} rel")
    assert scale > (5.0 if n == 32 else 30.0), "the weights are meant to be UNconditioned (here 1.0e1 / 7.4e1 / 5.6e1; SURVEY 7.3's draw: 3.8e1 / 4.1e2)"
    assert hip <= 4.0 * gap32 + 1e-7, (n, graphs, hip, gap32)


@pytest.mark.parametrize("n,graphs", [(32, 1), (64, 1), (64, 64)])
def test_unconditioned_sum_model_bf16_state_vs_fp64(n, graphs):
    scale, hip, gap32 = _run(n, graphs, "bf16")
    print(f"\n[dynamic range] N={n} x{graphs} bf16 state: max|logit| {scale:.3e}, HIP vs fp64 {hip:.3e} rel (fp32 oracle {gap32:.3e})")
    assert hip <= BF16_STATE_REL_BOUND, (n, graphs, hip)

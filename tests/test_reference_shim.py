"""The import path the reference's callers use -- `from models.mpn import MOTMPNet` (main.py:21, train.py:36, main_training.py:38) --
through the shipped shim `examples/reference_shim/models/mpn.py`, in a FRESH interpreter whose sys.path
holds the shim directory the way the reference's repository root holds `models/`.

CPU part: the name resolves to this package's class and the constructor / state_dict contract of SURVEY 8b holds under that name.
GPU part: the reference's own call sequence -- main.py:76-83 (`load_model_mpn`: construct, `.cuda()`, load weights), main.py:323-325
(`.cuda()`, `.eval()`), inference.py:283-291 (forward under no_grad, last classified step, sigmoid, >= 0.5) -- on a golden case, against
the logits the reference module itself produced (tests/golden/terrace32.npz)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN_DIR, ROOT

SHIM_DIR = os.path.join(ROOT, "examples", "reference_shim")

_CPU = r"""
import sys
sys.path.insert(0, {shim!r})          # what the reference's repository root is to main.py
from models.mpn import MOTMPNet, MetaLayer, EdgeModel, NodeModel, MLPGraphIndependent
import models.mpn as shim
import gnn_cca_amd.mpn as ours
assert MOTMPNet is ours.MOTMPNet and MetaLayer is ours.MetaLayer and MLPGraphIndependent is ours.MLPGraphIndependent
sys.path.insert(0, {root!r})
import copy
import numpy as np
from oracle.mpn_oracle import load_case
params, arch, sd, a = load_case({case!r})
p = copy.deepcopy(params)
m = MOTMPNet(p, None, arch)                                    # main.py:76
assert sorted(m.state_dict().keys()) == sorted(sd.keys())      # utils.load_pretrained_weights matches by name and size
assert all(tuple(m.state_dict()[k].shape) == tuple(np.asarray(v).shape) for k, v in sd.items())
try:
    bad = copy.deepcopy(params); bad['node_agg_fn'] = 'median'
    MOTMPNet(bad, None, arch)
    raise SystemExit('a bad node_agg_fn must assert (mpn.py:193)')
except AssertionError:
    pass
print('shim-cpu-ok')
"""

_GPU = r"""
import sys
sys.path.insert(0, {shim!r})
from models.mpn import MOTMPNet                                # main.py:21
sys.path.insert(0, {root!r})
import copy
import numpy as np
import torch
from oracle.mpn_oracle import load_case
params, arch, sd, a = load_case({case!r})
CONFIG = {{'GRAPH_NET_PARAMS': copy.deepcopy(params), 'CNN_MODEL': {{'arch': arch}}}}
model = MOTMPNet(CONFIG['GRAPH_NET_PARAMS'], None, CONFIG['CNN_MODEL']['arch']).cuda()     # main.py:76
model.load_state_dict({{k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, strict=True)   # utils.py:493
mpn_model = model
mpn_model.cuda()                                               # main.py:324
mpn_model.eval()                                               # main.py:325
class Batch: pass
data_batch = Batch()
data_batch.x, data_batch.edge_index, data_batch.edge_attr = (torch.from_numpy(a[k]).cuda() for k in ('x', 'edge_index', 'edge_attr'))
with torch.no_grad():                                          # inference.py:171
    outputs = mpn_model(data_batch)                            # inference.py:283
    preds = outputs['classified_edges'][-1].view(-1)           # inference.py:286
    sig = torch.nn.Sigmoid()
    preds_prob = sig(preds)
    predictions = (preds_prob >= 0.5) * 1
assert isinstance(outputs['classified_edges'], list) and len(outputs['classified_edges']) == 3
err = max(float(np.abs(o.view(-1).cpu().numpy() - a[f'logits_{{i}}'].reshape(-1)).max()) for i, o in enumerate(outputs['classified_edges']))
assert err <= 5e-6, err
ref_pred = (1.0 / (1.0 + np.exp(-a['logits_2'].reshape(-1).astype(np.float64))) >= 0.5) * 1
firm = np.abs(a['logits_2'].reshape(-1)) > 1e-4
assert np.array_equal(predictions.cpu().numpy()[firm], ref_pred[firm])
import gnn_cca_amd._native as nat
assert nat.lib() is not None
print('shim-gpu-ok', err)
"""


def _run(code):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)   # nothing but the shim directory may make `models.mpn` resolvable
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_models_mpn_resolves_to_this_package():
    out = _run(_CPU.format(shim=SHIM_DIR, root=ROOT, case=os.path.join(GOLDEN_DIR, "terrace32.npz")))
    assert "shim-cpu-ok" in out


@pytest.mark.gpu
def test_reference_call_sequence_through_the_shim():
    out = _run(_GPU.format(shim=SHIM_DIR, root=ROOT, case=os.path.join(GOLDEN_DIR, "terrace32.npz")))
    assert "shim-gpu-ok" in out

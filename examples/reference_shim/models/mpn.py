# models/mpn.py -- MI355X drop-in for the reference's module of the same path (vpulab/GNN-CCA models/mpn.py).
# A reference maintainer puts THIS file in place of models/mpn.py; main.py:21, train.py:36 and
# main_training.py:38 keep their `from models.mpn import MOTMPNet`.  GNNCCA_AMD_ROOT overrides the repository root.
import os
import sys

_ROOT = os.environ.get("GNNCCA_AMD_ROOT") or os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)   # repo root: contains gnn_cca_amd/ (import name) and gnn-cca_amd/ (sources + lib/)
from gnn_cca_amd.mpn import MOTMPNet, MetaLayer, EdgeModel, NodeModel, MLPGraphIndependent  # noqa: E402,F401

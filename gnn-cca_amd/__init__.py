"""MI355X-native message-passing path of GNN-CCA behind the reference's ``MOTMPNet`` interface.

    from gnn_cca_amd import MOTMPNet          # same constructor / forward / state_dict as models/mpn.py:144-299

The compute lives in ``lib/libgnncca_mpn.so`` (hand-written gfx950 HIP kernels behind the C ABI declared in
``include/gnncca_mpn.h``); this package is the Python host side that mirrors the reference's module surface.
"""
from .mlp import MLP  # noqa: F401
from .mpn import EdgeModel, MetaLayer, MLPGraphIndependent, MOTMPNet, NodeModel  # noqa: F401

__all__ = ["MOTMPNet", "MetaLayer", "EdgeModel", "NodeModel", "MLPGraphIndependent", "MLP"]

"""In-tree import name of the package directory ``gnn-cca_amd/`` (a hyphen cannot appear in a Python module name).

This is an ordinary package whose search path is extended to that directory: ``gnn_cca_amd.mpn``, ``gnn_cca_amd._native`` ...
resolve to the files under ``gnn-cca_amd/`` through the normal import machinery (no exec).  An installed copy (``setup.py`` maps
the package name onto the directory) does not need this file.

    from gnn_cca_amd import MOTMPNet          # same constructor / forward / state_dict as models/mpn.py:144-299
"""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gnn-cca_amd"))

from .mlp import MLP  # noqa: E402,F401
from .mpn import EdgeModel, MetaLayer, MLPGraphIndependent, MOTMPNet, NodeModel  # noqa: E402,F401

__all__ = ["MOTMPNet", "MetaLayer", "EdgeModel", "NodeModel", "MLPGraphIndependent", "MLP"]

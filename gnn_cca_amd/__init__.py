"""Importable alias of the package directory ``gnn-cca_amd/`` (a hyphen cannot appear in a Python module name).

``import gnn_cca_amd`` executes ``gnn-cca_amd/__init__.py`` with this package's ``__path__`` pointing there, so
``gnn_cca_amd.mpn``, ``gnn_cca_amd._native`` ... resolve to the files under ``gnn-cca_amd/``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gnn-cca_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

"""Installs the package directory ``gnn-cca_amd/`` under its import name ``gnn_cca_amd`` (with the prebuilt HIP library, if present:
``python gnn-cca_amd/build.py`` first).  In-tree use needs no install: the alias package ``gnn_cca_amd/`` extends its path there."""
from setuptools import setup

setup(
    name="gnn-cca-amd",
    version="0.4.0",
    description="MI355X-native message-passing path of GNN-CCA behind the reference's MOTMPNet interface",
    packages=["gnn_cca_amd"],
    package_dir={"gnn_cca_amd": "gnn-cca_amd"},
    package_data={"gnn_cca_amd": ["lib/*.so", "csrc/*"]},
    python_requires=">=3.10",
)

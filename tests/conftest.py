"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI library loads + exports (no GPU needed).
`-m gpu`       : parity tests proper -- HIP path through the C-ABI vs the oracle / golden vectors.
"""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_cases():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("weights_", "graph_", "ckpt_", "post_", "post2_", "bwd_", "drop_", "lw_", "rng_", "terrace_"))]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR

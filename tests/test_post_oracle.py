"""oracle/post_oracle.py (numpy / scipy restatement of inference.py:286-291 and libs/utils.py:295-317, 387-404) against the goldens the
reference's own functions produced (tests/golden/post_*.npz, make_golden_post.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import post_oracle as po

CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "post_*.npz")))


@pytest.mark.parametrize("name", CASES)
def test_post_oracle_matches_reference(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"post_{name}.npz"))
    n = int(z["n_nodes"])
    probs, preds = po.threshold(z["logits"])
    assert np.abs(probs - z["probs"]).max() <= 2e-7
    assert np.array_equal(preds, z["predictions"])
    pruned = po.prune(z["edge_index"], preds)
    assert np.array_equal(pruned, z["pruned"])
    fo, fi = po.flows(z["edge_index"], pruned, n)
    assert np.array_equal(fo, z["flow_out"]) and np.array_equal(fi, z["flow_in"])
    lab, k = po.clusters(z["edge_index"], pruned, n)
    assert k == int(z["n_clusters_pruned"]) and po.same_partition(lab, z["id_pruned"])
    lab, k = po.clusters(z["edge_index"], preds, n)
    assert k == int(z["n_clusters_raw"]) and po.same_partition(lab, z["id_raw"])

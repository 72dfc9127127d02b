"""Row N1: the CPU oracle of the graph-construction step against the golden vectors produced by the reference's own
statements, and the host-side plan (gnn_cca_amd.graph_build.plan_frames) against the same.  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import graph_oracle

CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "graph_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"graph_{name}.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    a = load(name)
    assert np.abs(graph_oracle.normalize_columns(a["reid_embeds_raw"]) - a["reid_embeds"]).max() <= 2.5e-7
    assert np.abs(graph_oracle.normalize_columns(a["node_embeds_raw"]) - a["x"]).max() <= 2.5e-7
    ei, attr, lab = graph_oracle.build(a["xw"], a["yw"], a["id"], a["id_cam"], a["graph_sizes"], a["max_dist"],
                                       a["reid_embeds"], bool(a["only_appearance"]), bool(a["only_dist"]))
    assert np.array_equal(ei, a["edge_index"])
    assert np.array_equal(lab, a["edge_labels"])
    assert attr.shape == a["edge_attr"].shape
    if not bool(a["only_appearance"]):
        assert np.array_equal(attr[:, :2], a["edge_attr"][:, :2]), "float64 ground-plane distances are bit-exact"
    assert np.abs(attr - a["edge_attr"]).max() <= 2e-6


@pytest.mark.parametrize("name", CASES)
def test_host_plan_enumerates_reference_edge_order(name):
    from gnn_cca_amd.graph_build import plan_frames
    a = load(name)
    plan = plan_frames(a["id_cam"], a["graph_sizes"])
    assert plan.n_edges == a["edge_index"].shape[1]
    # replay the kernel's enumeration on the host: source position p -> targets in ascending node id, other cameras
    rows, cols = [], []
    for p in range(len(a["id_cam"])):
        i = plan.src_order[p]
        g = plan.graph_of[i]
        tg = [j for j in range(plan.graph_ptr[g], plan.graph_ptr[g + 1]) if a["id_cam"][j] != a["id_cam"][i]]
        assert plan.edge_ptr[p + 1] - plan.edge_ptr[p] == len(tg)
        rows += [i] * len(tg)
        cols += tg
    assert np.array_equal(np.array([rows, cols]), a["edge_index"])

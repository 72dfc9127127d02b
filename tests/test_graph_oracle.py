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


def _native_plan(xw, yw, ids, id_cam, sizes, max_dist):
    """gnncca_plan_frames (a HOST function of the C-ABI library: no GPU involved) -> the fields of its staging image."""
    from gnn_cca_amd import _native as nat
    lib = nat.lib()
    xw, yw, md = (np.ascontiguousarray(v, np.float64) for v in (xw, yw, max_dist))
    ids, id_cam, sizes = (np.ascontiguousarray(v, np.int64) for v in (ids, id_cam, sizes))
    n, g = len(id_cam), len(sizes)
    nbytes = lib.gnncca_plan_frames_bytes(n, g)
    assert nbytes == 8 * (3 * n + g) + 4 * (5 * n + 2 * g + 3)
    buf = np.full(nbytes + 16, 0xAB, np.uint8)
    e = lib.gnncca_plan_frames(xw.ctypes.data, yw.ctypes.data, ids.ctypes.data, id_cam.ctypes.data, n, sizes.ctypes.data, md.ctypes.data, g,
                               buf.ctypes.data, nbytes)
    assert np.all(buf[nbytes:] == 0xAB), "wrote beyond the size it asked for"
    if e < 0:
        return int(e), None
    f64 = buf[:8 * (2 * n + g)].view(np.float64)
    i64 = buf[8 * (2 * n + g):8 * (3 * n + g)].view(np.int64)
    i32 = buf[8 * (3 * n + g):nbytes].view(np.int32)
    out = {"xw": f64[:n], "yw": f64[n:2 * n], "max_dist": f64[2 * n:], "ids": i64, "person": i32[:n], "cam": i32[n:2 * n],
           "graph_of": i32[2 * n:3 * n], "graph_ptr": i32[3 * n:3 * n + g + 1], "src_order": i32[3 * n + g + 1:4 * n + g + 1],
           "edge_ptr": i32[4 * n + g + 1:5 * n + g + 2], "edge_ptr_g": i32[5 * n + g + 2:5 * n + 2 * g + 3]}
    return int(e), out


def _check_native_plan(xw, yw, ids, id_cam, sizes, max_dist):
    from gnn_cca_amd.graph_build import plan_frames
    e, got = _native_plan(xw, yw, ids, id_cam, sizes, max_dist)
    want = plan_frames(id_cam, sizes)
    assert e == want.n_edges
    assert np.array_equal(got["src_order"], want.src_order) and np.array_equal(got["edge_ptr"], want.edge_ptr)
    assert np.array_equal(got["graph_ptr"], want.graph_ptr) and np.array_equal(got["graph_of"], want.graph_of)
    assert np.array_equal(got["edge_ptr_g"], want.edge_ptr[want.graph_ptr])
    assert np.array_equal(got["xw"], np.asarray(xw, np.float64)) and np.array_equal(got["yw"], np.asarray(yw, np.float64))
    assert np.array_equal(got["max_dist"], np.asarray(max_dist, np.float64)) and np.array_equal(got["ids"], np.asarray(ids, np.int64))
    assert np.array_equal(got["cam"], np.asarray(id_cam, np.int32))
    ids = np.asarray(ids)
    same = ids[:, None] == ids[None, :]
    assert np.array_equal(got["person"][:, None] == got["person"][None, :], same), "the relabelling must preserve equality of identities"


@pytest.mark.parametrize("name", CASES)
def test_native_host_plan_matches_the_numpy_plan_on_the_goldens(name):
    a = load(name)
    _check_native_plan(a["xw"], a["yw"], a["id"], a["id_cam"], a["graph_sizes"], a["max_dist"])


def test_native_host_plan_on_ragged_batches_and_its_refusals():
    from gnn_cca_amd import _native as nat
    rng = np.random.default_rng(4)
    for trial in range(30):
        g = int(rng.integers(1, 12))
        sizes = rng.integers(0, 40, size=g)            # empty frames included
        n = int(sizes.sum())
        cams = rng.choice([-3, 0, 1, 2, 7, 100000], size=n)   # camera ids need not be small or dense; one-camera frames give no edges
        ids = rng.choice([5, 9, 2 ** 40 + 1, -7, 123456789], size=n) if trial % 2 else rng.integers(0, 15, size=n)
        _check_native_plan(rng.normal(size=n), rng.normal(size=n), ids, cams, sizes, rng.uniform(1, 100, size=g))
    _check_native_plan([], [], [], [], [0, 0], [1.0, 2.0])
    # graph_sizes that do not sum to the number of detections
    e, _ = _native_plan([0.0] * 3, [0.0] * 3, [1, 2, 3], [0, 1, 0], [2, 2], [1.0, 1.0])
    assert e == -nat.ERR_INVALID_ARG
    e, _ = _native_plan([0.0] * 3, [0.0] * 3, [1, 2, 3], [0, 1, 0], [4, -1], [1.0, 1.0])
    assert e == -nat.ERR_INVALID_ARG
    # a camera id beyond int32 (the kernel keeps cameras as int32)
    e, _ = _native_plan([0.0] * 2, [0.0] * 2, [1, 2], [0, 2 ** 40], [2], [1.0])
    assert e == -nat.ERR_UNSUPPORTED

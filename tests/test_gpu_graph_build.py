"""Row N1 on the GPU: gnncca_build_edges / gnncca_normalize_columns (through gnn_cca_amd.graph_build) against the golden
vectors produced by the reference's own statements, then end to end into the MPN."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import graph_oracle

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "graph_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"graph_{name}.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_build_matches_reference(name):
    from gnn_cca_amd.graph_build import build_graph_batch
    a = load(name)
    b = build_graph_batch(a["xw"], a["yw"], a["id"], a["id_cam"], a["graph_sizes"], a["max_dist"],
                          torch.from_numpy(a["node_embeds_raw"]).cuda(), torch.from_numpy(a["reid_embeds_raw"]).cuda(),
                          only_appearance=bool(a["only_appearance"]), only_dist=bool(a["only_dist"]))
    torch.cuda.synchronize()
    assert np.abs(b.x.cpu().numpy() - a["x"]).max() <= 2.5e-7
    assert np.array_equal(b.edge_index.cpu().numpy(), a["edge_index"])
    assert np.array_equal(b.edge_labels.cpu().numpy(), a["edge_labels"])
    assert np.array_equal(b.y.cpu().numpy(), a["y"])
    attr = b.edge_attr.cpu().numpy()
    assert attr.shape == a["edge_attr"].shape
    if not bool(a["only_appearance"]):
        assert np.array_equal(attr[:, :2], a["edge_attr"][:, :2]), "float64 ground-plane distances must be bit-exact"
    assert np.abs(attr - a["edge_attr"]).max() <= 3e-6


def test_large_batch_vs_oracle():
    """512 frames x 4 cameras x 8 detections against the CPU oracle (sizes the golden files do not reach)."""
    from gnn_cca_amd.graph_build import build_graph_batch
    rng = np.random.default_rng(0)
    g, per = 512, 32
    n = g * per
    id_cam = np.tile(np.repeat(np.arange(4), 8), g)
    ids = rng.integers(0, 12, size=n)
    xw, yw = rng.uniform(-10, 10, n), rng.uniform(-10, 10, n)
    max_dist = rng.uniform(10, 90, g)
    reid = rng.standard_normal((n, 256)).astype(np.float32)
    b = build_graph_batch(xw, yw, ids, id_cam, [per] * g, max_dist, torch.zeros(n, 8).cuda(), torch.from_numpy(reid).cuda())
    reid_n = graph_oracle.normalize_columns(reid)
    ei, attr, lab = graph_oracle.build(xw, yw, ids, id_cam, [per] * g, max_dist, reid_n)
    assert np.array_equal(b.edge_index.cpu().numpy(), ei)
    assert np.array_equal(b.edge_labels.cpu().numpy(), lab)
    got = b.edge_attr.cpu().numpy()
    assert np.array_equal(got[:, :2], attr[:, :2])
    assert np.abs(got - attr).max() <= 3e-6


def test_end_to_end_into_mpn():
    """build_graph_batch -> MOTMPNet.forward, against oracle(graph) -> oracle(MPN)."""
    import copy

    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.graph_build import build_graph_batch
    from oracle.mpn_oracle import NumpyOracle, load_case
    a = load("terrace32")
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))  # node_in 64 weights
    rng = np.random.default_rng(5)
    node_raw = rng.standard_normal((32, 64)).astype(np.float32)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    b = build_graph_batch(a["xw"], a["yw"], a["id"], a["id_cam"], a["graph_sizes"], a["max_dist"],
                          torch.from_numpy(node_raw).cuda(), torch.from_numpy(a["reid_embeds_raw"]).cuda())
    with torch.no_grad():
        out = m(b)["classified_edges"]
    ref = NumpyOracle(params, arch, sd, np.float32).forward(graph_oracle.normalize_columns(node_raw), a["edge_index"],
                                                            a["edge_attr"])
    for o, r in zip(out, ref):
        assert np.abs(o.cpu().numpy() - r).max() <= 1e-5


# (rows <= 2304 with 16-byte aligned matrices of widths % 4 == 0: the LDS-resident kernel; 2304 / 2305 straddle its limit, 20 / 12 / 2052 leave a
#  workgroup's last column groups empty, 1 / 15 / 17 rows the DMA's 16-row groups)
@pytest.mark.parametrize("rows,cols,cols2", [(1229, 2048, 256), (37, 512, 0), (4096, 130, 7), (1, 5, 3), (65, 33, 2048), (2304, 2048, 256),
                                             (2305, 2048, 256), (63, 20, 12), (1100, 2052, 0), (1, 2048, 256), (15, 64, 16), (17, 64, 4),
                                             (129, 256, 0)])
def test_one_launch_normalisation_is_bitwise_the_three_kernel_form(rows, cols, cols2):
    """gnncca_normalize_columns2 (one launch, up to two matrices: what build_graph_batch uses for batches of <= 4096 detections) against
    gnncca_normalize_columns (three launches, any size) on the same matrices: the same bits, and the oracle within rounding."""
    from gnn_cca_amd import _native as nat
    from gnn_cca_amd.graph_build import normalize_columns
    g = torch.Generator().manual_seed(rows * 7 + cols)
    a = (torch.randn((rows, cols), generator=g) * 3).cuda()
    b = torch.rand((rows, cols2), generator=g).cuda() if cols2 else None
    if b is not None and cols2 > 2:
        b[:, 1] = 0.0                      # a zero column: the norm clamps at 1e-12, the column stays zero
    lib = nat.lib()

    def three(x):
        out = torch.empty_like(x)
        scratch = torch.empty(((rows + 63) // 64 + 1) * x.shape[1], dtype=torch.float32, device="cuda")
        nat.check(lib.gnncca_normalize_columns(x.data_ptr(), rows, x.shape[1], scratch.data_ptr(), out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "three-kernel form")
        return out
    if b is None:
        got = normalize_columns(a)
        assert torch.equal(got, three(a))
    else:
        got, got_b = normalize_columns(a, b)
        assert torch.equal(got, three(a)) and torch.equal(got_b, three(b))
        assert np.abs(got_b.cpu().numpy() - graph_oracle.normalize_columns(b.cpu().numpy())).max() <= 2.5e-7
    assert np.abs(got.cpu().numpy() - graph_oracle.normalize_columns(a.cpu().numpy())).max() <= 2.5e-7
    # an unaligned view (odd row offset of an odd-width matrix) takes the scalar path of both forms
    if cols % 2 == 1 and rows > 2:
        v = a.reshape(-1)[1:1 + (rows - 1) * cols].view(rows - 1, cols)
        assert v.data_ptr() % 16 != 0
        out = torch.empty((rows - 1, cols), device="cuda")
        nat.check(lib.gnncca_normalize_columns2(v.data_ptr(), cols, out.data_ptr(), None, 0, None, rows - 1,
                                                torch.cuda.current_stream().cuda_stream), "one-launch form")
        assert np.abs(out.cpu().numpy() - graph_oracle.normalize_columns(v.cpu().numpy())).max() <= 2.5e-7


def test_normalisation_beyond_4096_rows_takes_the_three_kernel_form():
    from gnn_cca_amd import _native as nat
    from gnn_cca_amd.graph_build import normalize_columns
    a = torch.randn((5000, 64), generator=torch.Generator().manual_seed(1)).cuda()
    b = torch.randn((5000, 16), generator=torch.Generator().manual_seed(2)).cuda()
    got, got_b = normalize_columns(a, b)
    assert np.abs(got.cpu().numpy() - graph_oracle.normalize_columns(a.cpu().numpy())).max() <= 2.5e-7
    assert np.abs(got_b.cpu().numpy() - graph_oracle.normalize_columns(b.cpu().numpy())).max() <= 2.5e-7
    out = torch.empty_like(a)
    st = nat.lib().gnncca_normalize_columns2(a.data_ptr(), 64, out.data_ptr(), None, 0, None, 5000, torch.cuda.current_stream().cuda_stream)
    assert st == nat.ERR_UNSUPPORTED


def test_many_batches_in_flight_reuse_the_pinned_staging_ring_safely():
    """build_graph_batch never synchronises: the staging image goes through a ring of eight pinned buffers and one non-blocking upload.
    Thirty batches of different sizes are enqueued back to back (the ring wraps three times, buffers grow on the way) with a long kernel
    queue in front, and every one must come out right -- a slot must not be rewritten before the copy that reads it has run."""
    from gnn_cca_amd.graph_build import build_graph_batch
    rng = np.random.default_rng(11)
    batches = []
    for i in range(30):
        g = int(rng.integers(1, 40)) if i != 17 else 300            # one batch far bigger than the others: its slot must grow
        sizes = rng.integers(0, 24, size=g)
        n = int(sizes.sum())
        if n == 0:
            sizes[0] = 5
            n = 5
        batches.append(dict(sizes=sizes, n=n, id_cam=rng.integers(0, 4, size=n), ids=rng.integers(0, 9, size=n), xw=rng.uniform(-10, 10, n),
                            yw=rng.uniform(-10, 10, n), max_dist=rng.uniform(10, 90, g), reid=rng.standard_normal((n, 64)).astype(np.float32)))
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):                                              # a queue of work in front: the uploads run late, the host runs ahead
        busy = busy @ busy * 1e-4
    outs = []
    for b in batches:
        outs.append(build_graph_batch(b["xw"], b["yw"], b["ids"], b["id_cam"], b["sizes"], b["max_dist"], torch.zeros(b["n"], 8).cuda(),
                                      torch.from_numpy(b["reid"]).cuda()))
    torch.cuda.synchronize()
    for b, o in zip(batches, outs):
        ei, attr, lab = graph_oracle.build(b["xw"], b["yw"], b["ids"], b["id_cam"], b["sizes"], b["max_dist"],
                                           graph_oracle.normalize_columns(b["reid"]))
        assert np.array_equal(o.edge_index.cpu().numpy(), ei)
        assert np.array_equal(o.edge_labels.cpu().numpy(), lab)
        assert np.array_equal(o.y.cpu().numpy(), np.asarray(b["ids"], np.int64))
        if ei.shape[1]:
            assert np.abs(o.edge_attr.cpu().numpy() - attr).max() <= 3e-6
        assert o.node_ptr == np.concatenate([[0], np.cumsum(b["sizes"])]).tolist()
        assert o.node_ptr_dev.cpu().tolist() == o.node_ptr and o.edge_ptr_dev.cpu().tolist() == o.edge_ptr

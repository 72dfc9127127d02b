"""Row N2 on the GPU against golden vectors produced by the reference's own utils functions."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "post_*.npz")))


def same_partition(a, b):
    return np.array_equal(a[:, None] == a[None, :], b[:, None] == b[None, :])


@pytest.mark.parametrize("name", CASES)
def test_threshold_prune_cluster(name):
    from gnn_cca_amd.postprocess import prune_and_cluster, threshold
    z = np.load(os.path.join(GOLDEN_DIR, f"post_{name}.npz"))
    n = int(z["n_nodes"])
    probs, preds = threshold(torch.from_numpy(z["logits"]).cuda().view(-1, 1))
    assert np.abs(probs.cpu().numpy() - z["probs"]).max() <= 2e-7
    assert np.array_equal(preds.cpu().numpy(), z["predictions"])
    out = prune_and_cluster(torch.from_numpy(z["edge_index"]).cuda(), preds, n)
    torch.cuda.synchronize()
    assert np.array_equal(out["pruned"].cpu().numpy(), z["pruned"])
    assert np.array_equal(out["flow_out"].cpu().numpy(), z["flow_out"])
    assert np.array_equal(out["flow_in"].cpu().numpy(), z["flow_in"])
    assert int(out["n_clusters"].item()) == int(z["n_clusters_pruned"])
    assert same_partition(out["labels"].cpu().numpy(), z["id_pruned"])  # cluster ids are arbitrary, the partition is not


def test_cluster_large_batch_property():
    """256 frames of 4 x 8 detections: planted identities with symmetric active edges -> the clusters are the identities."""
    from gnn_cca_amd.postprocess import prune_and_cluster
    rng = np.random.default_rng(2)
    frames, per = 256, 32
    n = frames * per
    cam = np.tile(np.repeat(np.arange(4), 8), frames)
    ident = np.concatenate([f * 100 + rng.integers(0, 10, per) for f in range(frames)])
    rows, cols = [], []
    for f in range(frames):
        idx = np.arange(f * per, (f + 1) * per)
        i, j = np.meshgrid(idx, idx, indexing="ij")
        m = cam[i] != cam[j]
        rows.append(i[m])
        cols.append(j[m])
    ei = np.stack([np.concatenate(rows), np.concatenate(cols)])
    pred = (ident[ei[0]] == ident[ei[1]]).astype(np.int64)
    extra = rng.random(ei.shape[1]) < 0.02          # spurious one-directional activations: pruning must remove them
    lower = ei[0] < ei[1]
    pred_noisy = pred | (extra & lower & (pred == 0))
    out = prune_and_cluster(torch.from_numpy(ei).cuda(), torch.from_numpy(pred_noisy).cuda(), n)
    assert np.array_equal(out["pruned"].cpu().numpy(), pred)
    lab = out["labels"].cpu().numpy()
    # two detections share a cluster iff they are connected through same-identity cross-camera pairs
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    act = pred == 1  # csgraph looks at the sparsity structure: pass the active edges only
    a = sp.coo_matrix((np.ones(int(act.sum())), (ei[0][act], ei[1][act])), shape=(n, n))
    k, ref = connected_components(a, directed=False)
    assert int(out["n_clusters"].item()) == k
    assert np.array_equal(lab == lab[0], ref == ref[0]) and len(np.unique(lab)) == k
    _, inv = np.unique(lab, return_inverse=True)
    _, inv_ref = np.unique(ref, return_inverse=True)
    first = {}
    assert all(first.setdefault(a_, b_) == b_ for a_, b_ in zip(inv, inv_ref))


def test_frames_mode_equals_single_workgroup():
    """Per-frame clustering (one workgroup per frame, node_ptr / edge_ptr ranges) gives exactly the labels, pruning,
    flows and cluster count of the whole-batch form -- on a batch built by build_graph_batch (device ranges) and with
    host lists."""
    import numpy as np
    from gnn_cca_amd.graph_build import build_graph_batch
    from gnn_cca_amd.postprocess import prune_and_cluster
    rng = np.random.default_rng(5)
    frames, cams, per = 37, 4, 6
    n_g = cams * per
    n = frames * n_g
    id_cam = np.tile(np.repeat(np.arange(cams), per), frames)
    ids = rng.integers(0, per, size=n).astype(np.int64)
    xw, yw = rng.normal(size=n), rng.normal(size=n)
    node = torch.randn(n, 64, device="cuda")
    reid = torch.randn(n, 32, device="cuda")
    batch = build_graph_batch(xw, yw, ids, id_cam, [n_g] * frames, [50.0] * frames, node, reid)
    e = batch.edge_index.shape[1]
    preds = (torch.rand(e, device="cuda") < 0.35).long()
    ref = prune_and_cluster(batch.edge_index, preds, n)
    for np_, ep_ in ((batch.node_ptr_dev, batch.edge_ptr_dev), (batch.node_ptr, batch.edge_ptr)):
        out = prune_and_cluster(batch.edge_index, preds, n, node_ptr=np_, edge_ptr=ep_)
        for k in ("pruned", "flow_out", "flow_in", "labels", "n_clusters"):
            assert torch.equal(out[k], ref[k]), k
    assert int(ref["n_clusters"]) >= frames


def test_separate_counter_buffers_and_empty_inputs():
    """The C entry zeroes flow_out / flow_in / n_clusters itself whether or not the caller laid them out back to back (one memset or
    three), on buffers that held garbage; N = 0 gives zero clusters."""
    import ctypes as C

    from gnn_cca_amd import _native as nat
    from gnn_cca_amd.postprocess import prune_and_cluster
    from oracle import post_oracle as po
    rng = np.random.default_rng(3)
    n = 50
    ei = np.array([(i, j) for i in range(n) for j in range(n) if i != j and (i + j) % 3], dtype=np.int64).T.copy()
    pred = (rng.random(ei.shape[1]) < 0.4).astype(np.int64)
    want_pruned = po.prune(ei, pred)
    lab, k = po.clusters(ei, want_pruned, n)
    fo = np.bincount(ei[0][want_pruned == 1], minlength=n)
    fi = np.bincount(ei[1][want_pruned == 1], minlength=n)
    d_ei, d_pred = torch.from_numpy(ei).cuda(), torch.from_numpy(pred).cuda()
    out = prune_and_cluster(d_ei, d_pred, n)                    # the module's layout: one block, one memset
    assert np.array_equal(out["flow_out"].cpu().numpy(), fo) and np.array_equal(out["flow_in"].cpu().numpy(), fi)
    assert int(out["n_clusters"].item()) == k and np.array_equal(out["pruned"].cpu().numpy(), want_pruned)
    lib = nat.lib()
    e = ei.shape[1]
    ws = torch.empty(lib.gnncca_post_workspace_bytes(n, e) + 256, dtype=torch.uint8, device="cuda")
    bufs = [torch.full((m,), 12345, dtype=torch.int32, device="cuda") for m in (n, n, n, 1)]    # separate, dirty buffers
    pruned = torch.empty(e, dtype=torch.int64, device="cuda")
    st = lib.gnncca_post_prune_cluster_frames(d_ei.data_ptr(), d_pred.data_ptr(), n, e, None, None, 0, ws.data_ptr(), ws.numel(),
                                              pruned.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
                                              bufs[3].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert st == 0
    assert np.array_equal(bufs[0].cpu().numpy(), fo) and np.array_equal(bufs[1].cpu().numpy(), fi) and int(bufs[3].item()) == k
    assert po.same_partition(bufs[2].cpu().numpy(), lab)
    # no nodes at all
    out = prune_and_cluster(torch.zeros((2, 0), dtype=torch.int64, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"), 0)
    assert int(out["n_clusters"].item()) == 0 and out["labels"].numel() == 0
    # nodes without edges: every node its own cluster, zero flows
    out = prune_and_cluster(torch.zeros((2, 0), dtype=torch.int64, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"), 7)
    assert int(out["n_clusters"].item()) == 7 and int(out["flow_out"].abs().sum()) == 0 and int(out["flow_in"].abs().sum()) == 0

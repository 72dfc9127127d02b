"""gnn_cca_amd.pipeline.FramePipeline: a batch of frames from detections to identity clusters in ONE native call
(gnncca_frames_forward) -- the launches of graph_build.build_graph_batch, MOTMPNet.forward, postprocess.threshold and
postprocess.prune_and_cluster issued without the Python in between.  Every output must be BIT FOR BIT the step-by-step path's
(same kernels, same arguments), which the other GPU tests pin against the oracles and the reference's goldens."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(rng, g, lo=0, hi=24, cams=4):
    sizes = rng.integers(lo, hi, size=g)
    if sizes.sum() == 0:
        sizes[0] = 6
    n = int(sizes.sum())
    return dict(sizes=sizes, n=n, id_cam=rng.integers(0, cams, size=n), ids=rng.integers(0, 11, size=n), xw=rng.uniform(-10, 10, n),
                yw=rng.uniform(-10, 10, n), max_dist=rng.uniform(10, 90, g), node=rng.standard_normal((n, 2048)).astype(np.float32),
                reid=rng.standard_normal((n, 256)).astype(np.float32))


def _model(seed=0):
    import bench
    m = bench.build_model(copy.deepcopy(bench.graph_net_params(L=4)), 20, seed=seed).cuda().eval()
    return m


def _stepwise(m, f, node, reid):
    from gnn_cca_amd.graph_build import build_graph_batch
    from gnn_cca_amd.postprocess import prune_and_cluster, threshold
    b = build_graph_batch(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
    with torch.no_grad():
        out = m(b)
    probs, preds = threshold(out["classified_edges"][-1])
    post = prune_and_cluster(b.edge_index, preds, b.x.shape[0], b.node_ptr_dev, b.edge_ptr_dev)
    return b, out, probs, preds, post


def _same(r, ref):
    b, out, probs, preds, post = ref
    assert torch.equal(r.batch.x, b.x) and torch.equal(r.batch.edge_index, b.edge_index) and torch.equal(r.batch.edge_attr, b.edge_attr)
    assert torch.equal(r.batch.edge_labels, b.edge_labels) and torch.equal(r.batch.y, b.y) and torch.equal(r.batch.reid_embeds, b.reid_embeds)
    assert r.batch.node_ptr == b.node_ptr and r.batch.edge_ptr == b.edge_ptr
    assert torch.equal(r.batch.node_ptr_dev, b.node_ptr_dev) and torch.equal(r.batch.edge_ptr_dev, b.edge_ptr_dev)
    assert len(r.outputs["classified_edges"]) == len(out["classified_edges"])
    for a, c in zip(r.outputs["classified_edges"], out["classified_edges"]):
        assert a.shape == c.shape and torch.equal(a, c)
    assert torch.equal(r.probs, probs) and torch.equal(r.preds, preds)
    for k in ("pruned", "flow_out", "flow_in", "labels", "n_clusters", "triggers"):
        assert torch.equal(getattr(r, k), post[k]), k


@pytest.mark.parametrize("g,seed", [(64, 1), (1, 2), (7, 3), (200, 4)])
def test_one_call_equals_the_step_by_step_path(g, seed):
    from gnn_cca_amd.pipeline import FramePipeline
    rng = np.random.default_rng(seed)
    f = _frames(rng, g, hi=24 if g < 200 else 20)
    m = _model()
    node, reid = torch.from_numpy(f["node"]).cuda(), torch.from_numpy(f["reid"]).cuda()
    # put the decision boundary inside the logits so that pruning and clustering have work to do
    ref = _stepwise(m, f, node, reid)
    with torch.no_grad():
        sd = m.state_dict()
        key = [k for k in sd if k.startswith("classifier.") and k.endswith(".bias")][-1]
        sd[key] -= ref[1]["classified_edges"][-1].median()
        m.load_state_dict(sd)
    ref = _stepwise(m, f, node, reid)
    pipe = FramePipeline(m)
    for _ in range(3):                       # repeated calls: staging ring, workspace reuse
        r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
    torch.cuda.synchronize()
    _same(r, ref)
    assert int(r.n_clusters.item()) >= 1 and 0 < int(r.pruned.sum().item()) < r.pruned.numel()
    # the reference's FINAL result (ROUNDING / PRUNING / SPLITTING = True, inference.py:306-345): every frame through the oracle's restatement
    # of the heuristics (pinned by the reference's goldens in tests/test_post_heuristics.py) on the GPU's own probabilities
    from oracle import post_oracle as po
    fin = r.final()
    assert fin is r.final()
    ei, probs = r.batch.edge_index.cpu().numpy(), r.probs.cpu().numpy()
    got_pred, got_lab = fin["predictions"].cpu().numpy(), fin["labels"].cpu().numpy()
    trig = r.triggers.cpu().numpy()
    total, flagged = 0, 0
    for q in range(len(r.batch.node_ptr) - 1):
        v0, v1, k0, k1 = r.batch.node_ptr[q], r.batch.node_ptr[q + 1], r.batch.edge_ptr[q], r.batch.edge_ptr[q + 1]
        _, want, ids, k = po.finalize(ei[:, k0:k1] - v0, None, v1 - v0, probs=probs[k0:k1])
        assert np.array_equal(got_pred[k0:k1], want), q
        assert po.same_partition(got_lab[v0:v1], ids), q
        total += k
        changed = not np.array_equal(want, r.pruned.cpu().numpy()[k0:k1])
        assert not (changed and trig[q] == 0), q          # a frame the heuristics change always raises a trigger
        flagged += int(trig[q] != 0)
    assert int(fin["n_clusters"].item()) == total
    assert fin["frames_finalized"] == [q for q in range(len(trig)) if trig[q]]
    if g >= 64:
        assert flagged >= 1


def test_shapes_alternate_and_fallbacks_agree():
    """Batches of different sizes through ONE pipeline object (workspace sizes are cached per shape); a batch without any cross-camera
    pair, a batch beyond 4096 detections and a hooked model take the step-by-step path inside the same call."""
    from gnn_cca_amd.pipeline import FramePipeline
    rng = np.random.default_rng(9)
    m = _model(seed=1)
    pipe = FramePipeline(m)
    for g in (30, 3, 90, 30):
        f = _frames(rng, g)
        node, reid = torch.from_numpy(f["node"]).cuda(), torch.from_numpy(f["reid"]).cuda()
        r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
        _same(r, _stepwise(m, f, node, reid))
    # one camera only: no edges
    f = _frames(rng, 5, lo=3, hi=9, cams=1)
    node, reid = torch.from_numpy(f["node"]).cuda(), torch.from_numpy(f["reid"]).cuda()
    r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
    assert r.batch.edge_index.shape == (2, 0) and int(r.n_clusters.item()) == f["n"]
    # beyond the one-launch normalisation's 4096 rows
    f = _frames(rng, 260, lo=14, hi=20)
    assert f["n"] > 4096
    node, reid = torch.from_numpy(f["node"]).cuda(), torch.from_numpy(f["reid"]).cuda()
    r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
    _same(r, _stepwise(m, f, node, reid))
    # refusals
    with pytest.raises(ValueError):
        pipe(f["xw"][:-1], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
    with pytest.raises(RuntimeError):
        pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node.cpu(), reid)


def test_final_async_overlaps_batches_and_equals_the_synchronous_finalize():
    """FrameResult.final_async (round 6): several batches in flight -- each handed to the pool of host threads behind ONE D2H copy behind the
    chain, no synchronisation in between -- collected OUT OF ORDER; every result equals postprocess.finalize (the synchronous
    host pass, pinned against the reference's goldens) on the same device tensors, and FrameResult.final() returns the same as device tensors.
    A result of the step-by-step path (more than 4096 detections) goes through the same interface."""
    from gnn_cca_amd.pipeline import FramePipeline
    from gnn_cca_amd.postprocess import finalize
    rng = np.random.default_rng(77)
    m = _model(seed=2)
    pipe = FramePipeline(m)
    pipe.host_threads = 4
    fs = [_frames(rng, g) for g in (64, 20, 64, 5, 90)] + [_frames(rng, 260, lo=14, hi=20)]
    dev_in = [(torch.from_numpy(f["node"]).cuda(), torch.from_numpy(f["reid"]).cuda()) for f in fs]
    # centre the logits so that the heuristics have work to do
    r0 = pipe(fs[0]["xw"], fs[0]["yw"], fs[0]["ids"], fs[0]["id_cam"], fs[0]["sizes"], fs[0]["max_dist"], *dev_in[0])
    with torch.no_grad():
        sd = m.state_dict()
        key = [k for k in sd if k.startswith("classifier.") and k.endswith(".bias")][-1]
        sd[key] -= r0.outputs["classified_edges"][-1].median()
        m.load_state_dict(sd)
    results, pending = [], []
    for f, (node, reid) in zip(fs, dev_in):                    # no synchronisation anywhere in this loop
        r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
        results.append(r)
        pending.append(r.final_async())
    assert pending[0] is results[0].final_async()
    touched = 0
    for i in (3, 0, 5, 4, 1, 2):
        r, got = results[i], pending[i].result()
        b = r.batch
        want = finalize(b.edge_index, r.probs, r.pruned, r.labels, r.n_clusters, r.triggers, b.node_ptr, b.edge_ptr)
        assert np.array_equal(got["predictions"], want["predictions"].cpu().numpy()), i
        assert np.array_equal(got["labels"], want["labels"].cpu().numpy()), i
        assert got["n_clusters"] == int(want["n_clusters"].item()) and got["frames_finalized"] == want["frames_finalized"], i
        assert np.array_equal(got["triggers"], r.triggers.cpu().numpy())
        fin = r.final()
        assert torch.equal(fin["predictions"], want["predictions"]) and torch.equal(fin["labels"], want["labels"])
        assert int(fin["n_clusters"].item()) == got["n_clusters"]
        touched += len(got["frames_finalized"])
        # the device chain's own tensors are untouched by the host pass
        assert torch.equal(r.pruned, want["predictions"]) == (not got["frames_finalized"] or torch.equal(r.pruned, want["predictions"]))
    assert touched >= 10
    # close() refuses while a batch is still with the pool, and works once it has been collected
    late = pipe(fs[1]["xw"], fs[1]["yw"], fs[1]["ids"], fs[1]["id_cam"], fs[1]["sizes"], fs[1]["max_dist"], *dev_in[1]).final_async()
    with pytest.raises(RuntimeError):
        pipe.close()
    late.result()
    del pending, results, r, got, fin, late
    pipe.close()


@pytest.mark.parametrize("route", ["generic_fused", "generic_op_by_op", "steps_L0", "max_aggregation", "reattach_both"])
def test_one_call_equals_the_step_by_step_path_on_every_forward_route(route):
    """gnncca_frames_forward hands the pruning the CSR plan the MPN forward left in ITS workspace (seg_ptr / col32 / perm / flags), found by
    carving the workspace again -- an invariant every forward route has to keep (csrc/pack.cpp: carve / carve_generic).  One pipeline call
    against the separate calls, bit for bit, on the routes beside the shipped shape's: the generic family's fused step (node latent 48) and
    its op-by-op form (node latent 160), L = 0 (no step kernel at all), the general step kernel (max aggregation; both reattach flags)."""
    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.pipeline import FramePipeline
    import bench
    params = copy.deepcopy(bench.graph_net_params(L=4))
    if route == "generic_fused":
        params["encoder_feats_dict"]["nodes"]["resnet50"]["node_out_dim"] = 48
        params["node_model_feats_dict"]["fc_dims"] = [48]
    elif route == "generic_op_by_op":
        params["encoder_feats_dict"]["nodes"]["resnet50"]["node_out_dim"] = 160
        params["node_model_feats_dict"]["fc_dims"] = [160]
    elif route == "steps_L0":
        params["num_enc_steps"], params["num_class_steps"] = 0, 1
    elif route == "max_aggregation":
        params["node_agg_fn"] = "max"
    elif route == "reattach_both":
        params["reattach_initial_nodes"] = params["reattach_initial_edges"] = True
    torch.manual_seed(3)
    m = MOTMPNet(copy.deepcopy(params), None, "resnet50")
    with torch.no_grad():
        for p in m.MPNet.node_model.node_mlp.parameters():
            p.mul_(1.0 / 20)
    m = m.cuda().eval()
    rng = np.random.default_rng(5)
    f = _frames(rng, 24)
    node, reid = torch.from_numpy(f["node"]).cuda(), torch.from_numpy(f["reid"]).cuda()
    ref = _stepwise(m, f, node, reid)
    with torch.no_grad():   # centre the logits: pruning and clustering get work to do
        sd = m.state_dict()
        key = [k for k in sd if k.startswith("classifier.") and k.endswith(".bias")][-1]
        sd[key] -= ref[1]["classified_edges"][-1].median()
        m.load_state_dict(sd)
    ref = _stepwise(m, f, node, reid)
    pipe = FramePipeline(m)
    for _ in range(2):
        r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
    torch.cuda.synchronize()
    _same(r, ref)
    assert 0 < int(r.pruned.sum().item()) < r.pruned.numel()

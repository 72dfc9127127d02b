"""SURVEY.md 8f row N2, second half: the reference's bridge-based heuristics -- degree-constraint rounding (libs/utils.py:25-173) and
big-cluster splitting (libs/utils.py:319-386) in the order of the shipped configuration (inference.py:306-345; config_inference.yaml:6-8).

  * the oracle (oracle/post_oracle.py: finalize) against tests/golden/post2_heuristics.npz, produced by the reference's own functions
    executed through inference.py's own lines (tests/golden/make_golden_post2.py): final predictions AND the reference's ID_pred, label
    numbering included, for the four switch settings stored;
  * the product's host implementation (csrc/post_host.cpp: gnncca_post_finalize_frame_host -- plain C++ in the library, no GPU involved)
    against the same goldens, and against the oracle on random frames (random camera layouts, noisy identity structure, duplicate
    probabilities) where the goldens do not reach;
  * [gpu] the device chain's trigger bits and `postprocess.finalize` end to end."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import post_oracle as po

Z = np.load(os.path.join(GOLDEN_DIR, "post2_heuristics.npz"))
NAMES = [str(n) for n in Z["names"]]
SETTINGS = {"final": (True, True, True), "rounded": (True, True, False), "pruned": (False, True, False), "split_only": (False, True, True)}


def case(name):
    g = lambda k: Z[f"{name}::{k}"]   # noqa: E731
    return int(g("n_nodes")), g("edge_index"), g("logits"), g("probs"), g


def native_finalize(n, ei, probs, pred_in, switches, base=0):
    from gnn_cca_amd import _native as nat
    src, dst = np.ascontiguousarray(ei[0] + base), np.ascontiguousarray(ei[1] + base)
    pr = np.ascontiguousarray(probs, dtype=np.float32)
    pred = np.ascontiguousarray(pred_in, dtype=np.int64).copy()
    labels, ids, k = np.zeros(n, np.int32), np.zeros(n, np.int64), C.c_int32(0)
    sw = (nat.POST_ROUNDING if switches[0] else 0) | (nat.POST_PRUNING if switches[1] else 0) | (nat.POST_SPLITTING if switches[2] else 0)
    st = nat.lib().gnncca_post_finalize_frame_host(src.ctypes.data, dst.ctypes.data, base, n, ei.shape[1], pr.ctypes.data, pred.ctypes.data, sw,
                                                   labels.ctypes.data, C.byref(k), ids.ctypes.data)
    assert st == 0
    return pred, ids, labels, int(k.value)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference(name):
    n, ei, logits, probs, g = case(name)
    assert np.abs(po.threshold(logits)[0] - probs).max() <= 2e-7      # the sigmoid itself: an ulp of float32
    for tag, sw in SETTINGS.items():
        _, pred, ids, k = po.finalize(ei, logits, n, *sw, probs=probs)
        assert np.array_equal(pred, g("pred_" + tag)), (name, tag)
        assert np.array_equal(ids, g("id_" + tag)), (name, tag)        # the reference's label numbering too
        assert k == len(set(g("id_" + tag).tolist()))


@pytest.mark.parametrize("name", NAMES)
def test_host_implementation_matches_the_reference(name):
    n, ei, logits, probs, g = case(name)
    thresholded = (probs >= 0.5).astype(np.int64)
    for tag, sw in SETTINGS.items():
        for base in (0, 1000):                                          # frame-local and batch-global node numbering
            pred, ids, labels, k = native_finalize(n, ei, probs, thresholded, sw, base)
            assert np.array_equal(pred, g("pred_" + tag)), (name, tag)
            assert np.array_equal(ids, g("id_" + tag)), (name, tag)
            assert k == len(set(ids.tolist())) and po.same_partition(labels, ids)
            assert all(labels[v] == base + min(u for u in range(n) if ids[u] == ids[v]) for v in range(n))
    # handing over the device chain's PRUNED predictions instead of the thresholded ones gives the same result (pruning is idempotent)
    pred2, ids2, _, _ = native_finalize(n, ei, probs, g("pred_pruned"), SETTINGS["final"])
    assert np.array_equal(pred2, g("pred_final")) and np.array_equal(ids2, g("id_final"))


def random_frame(rng):
    cams = rng.integers(1, 5, size=int(rng.integers(2, 7)))
    cam_of = np.repeat(np.arange(len(cams)), cams)
    n = len(cam_of)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = cam_of[i] != cam_of[j]
    ei = np.stack([i[m], j[m]]).astype(np.int64)
    ident = rng.integers(0, max(n // int(rng.integers(2, 6)), 1) + 1, size=n)
    same = ident[ei[0]] == ident[ei[1]]
    logits = np.where(same, rng.uniform(0.5, 3.0), -rng.uniform(1.0, 3.0)) + rng.normal(0, rng.uniform(0.3, 2.0), size=ei.shape[1])
    if rng.random() < 0.3:
        logits = np.round(logits * 2) / 2           # ties: many edges share a probability (splitting removes ALL of them)
    return n, ei, logits.astype(np.float32)


def test_host_implementation_matches_the_oracle_on_random_frames():
    rng = np.random.default_rng(2024)
    changed = 0
    for it in range(300):
        n, ei, logits = random_frame(rng)
        if ei.shape[1] == 0:
            continue
        probs, thresholded = po.threshold(logits)
        for sw in ((True, True, True), (False, True, True), (True, True, False)):
            _, want, want_ids, want_k = po.finalize(ei, logits, n, *sw, probs=probs)
            pred, ids, _, k = native_finalize(n, ei, probs, thresholded, sw)
            assert np.array_equal(pred, want) and np.array_equal(ids, want_ids) and k == want_k, (it, sw)
        changed += int(not np.array_equal(want, po.prune(ei, thresholded)))
    assert changed >= 50     # the heuristics had work to do on a good share of the frames


def test_threaded_batch_entry_point_matches_the_reference():
    """gnncca_post_finalize_frames_host: all golden frames as one Batch.from_data_list-style batch, every frame listed, dealt to host threads."""
    from gnn_cca_amd import _native as nat
    eis, probs, node_ptr, edge_ptr, want_pred, want_k = [], [], [0], [0], [], []
    for name in NAMES:
        n, ei, _, p, g = case(name)
        eis.append(ei + node_ptr[-1])
        probs.append(p)
        want_pred.append(g("pred_final"))
        want_k.append(len(set(g("id_final").tolist())))
        node_ptr.append(node_ptr[-1] + n)
        edge_ptr.append(edge_ptr[-1] + ei.shape[1])
    ei_all, pr = np.concatenate(eis, axis=1), np.ascontiguousarray(np.concatenate(probs), dtype=np.float32)
    src, dst = np.ascontiguousarray(ei_all[0]), np.ascontiguousarray(ei_all[1])
    np_h, ep_h = np.asarray(node_ptr, np.int32), np.asarray(edge_ptr, np.int32)
    for threads in (1, 4, 0):
        pred = (pr >= 0.5).astype(np.int64)
        labels = np.zeros(node_ptr[-1], np.int32)
        listed, k = np.arange(len(NAMES), dtype=np.int32)[::-1].copy(), np.zeros(len(NAMES), np.int32)
        st = nat.lib().gnncca_post_finalize_frames_host(src.ctypes.data, dst.ctypes.data, np_h.ctypes.data, ep_h.ctypes.data, listed.ctypes.data,
                                                        len(NAMES), pr.ctypes.data, pred.ctypes.data, 7, labels.ctypes.data, k.ctypes.data, threads)
        assert st == 0
        assert np.array_equal(pred, np.concatenate(want_pred))
        assert k[::-1].tolist() == want_k
        for q, name in enumerate(NAMES):
            assert po.same_partition(labels[node_ptr[q]:node_ptr[q + 1]], case(name)[4]("id_final")), name
            assert labels[node_ptr[q]:node_ptr[q + 1]].min() >= node_ptr[q]


def test_asynchronous_pool_matches_the_reference():
    """gnncca_post_pool_* (round 6): the persistent pool of host threads.  All golden frames as one batch, submitted several times (jobs in
    flight together, collected out of order, with and without a SPLITTING switch); the pool lists the flagged frames from the trigger words
    itself, finalizes exactly those, and returns the reference's final predictions / partition / cluster count.  Host data, no event."""
    from gnn_cca_amd import _native as nat
    lib = nat.lib()
    eis, probs, node_ptr, edge_ptr = [], [], [0], [0]
    for name in NAMES:
        n, ei, _, p, g = case(name)
        eis.append(ei + node_ptr[-1]), probs.append(p)
        node_ptr.append(node_ptr[-1] + n), edge_ptr.append(edge_ptr[-1] + ei.shape[1])
    ei_all, pr = np.concatenate(eis, axis=1), np.ascontiguousarray(np.concatenate(probs), dtype=np.float32)
    src, dst = np.ascontiguousarray(ei_all[0]), np.ascontiguousarray(ei_all[1])
    np_h, ep_h = np.asarray(node_ptr, np.int32), np.asarray(edge_ptr, np.int32)
    G, N = len(NAMES), node_ptr[-1]
    # what the device chain hands over: pruned predictions, labels = smallest node id of the component, its cluster count, the trigger words
    pruned = np.concatenate([case(nm)[4]("pred_pruned") for nm in NAMES]).astype(np.int64)
    labels0, k0, trig = np.zeros(N, np.int32), 0, np.zeros(G, np.int32)
    for q, nm in enumerate(NAMES):
        n, ei, _, _, g = case(nm)
        ids = g("id_pruned")
        for v in range(n):
            labels0[node_ptr[q] + v] = node_ptr[q] + min(u for u in range(n) if ids[u] == ids[v])
        k0 += len(set(ids.tolist()))
        fo, fi = po.flows(ei, g("pred_pruned"), n)
        trig[q] = (1 if (fo > 3).any() or (fi > 3).any() else 0) | (2 if np.bincount(ids).max() > 4 else 0)
    pool = lib.gnncca_post_pool_create(3)
    assert pool and lib.gnncca_post_pool_threads(pool) == 3
    try:
        jobs = []
        for rep, (tag, sw) in enumerate([("final", 7), ("rounded", 3), ("final", 7), ("split_only", 6), ("final", 7)]):
            pred, lab, kk = pruned.copy(), labels0.copy(), np.asarray([k0], np.int32)
            b = nat.PostBatch()
            b.src, b.dst, b.node_ptr, b.edge_ptr, b.n_frames, b.switches = src.ctypes.data, dst.ctypes.data, np_h.ctypes.data, ep_h.ctypes.data, G, sw
            b.triggers, b.probs, b.predictions, b.labels, b.n_clusters = trig.ctypes.data, pr.ctypes.data, pred.ctypes.data, lab.ctypes.data, kk.ctypes.data
            b.ready_event, b.device = None, 0
            t = lib.gnncca_post_pool_submit(pool, C.byref(b))
            assert t >= 0
            jobs.append((t, tag, sw, pred, lab, kk, b))
        for t, tag, sw, pred, lab, kk, _b in reversed(jobs):
            frames, cnt = np.zeros(G, np.int32), C.c_int32(0)
            assert lib.gnncca_post_pool_wait(pool, t, frames.ctypes.data, C.byref(cnt)) == 0
            want_mask = (1 if sw & 1 else 0) | (2 if sw & 4 else 0)
            assert frames[:cnt.value].tolist() == [q for q in range(G) if trig[q] & want_mask]
            assert np.array_equal(pred, np.concatenate([case(nm)[4]("pred_" + tag) for nm in NAMES])), tag
            want_ids = np.concatenate([case(nm)[4]("id_" + tag) + 1000 * q for q, nm in enumerate(NAMES)])
            assert po.same_partition(lab, want_ids), tag
            assert int(kk[0]) == len(set(want_ids.tolist())), tag
            assert lib.gnncca_post_pool_wait(pool, t, None, None) == nat.ERR_INVALID_ARG      # a ticket is collected once
        # a batch nobody flagged: final at once, nothing touched
        pred, lab, kk, none = pruned.copy(), labels0.copy(), np.asarray([k0], np.int32), np.zeros(G, np.int32)
        b = nat.PostBatch()
        b.src, b.dst, b.node_ptr, b.edge_ptr, b.n_frames, b.switches = src.ctypes.data, dst.ctypes.data, np_h.ctypes.data, ep_h.ctypes.data, G, 7
        b.triggers, b.probs, b.predictions, b.labels, b.n_clusters = none.ctypes.data, pr.ctypes.data, pred.ctypes.data, lab.ctypes.data, kk.ctypes.data
        t = lib.gnncca_post_pool_submit(pool, C.byref(b))
        cnt = C.c_int32(-1)
        assert lib.gnncca_post_pool_wait(pool, t, None, C.byref(cnt)) == 0 and cnt.value == 0
        assert np.array_equal(pred, pruned) and np.array_equal(lab, labels0) and int(kk[0]) == k0
        assert lib.gnncca_post_pool_submit(pool, None) == -nat.ERR_INVALID_ARG
    finally:
        lib.gnncca_post_pool_destroy(pool)


@pytest.mark.gpu
def test_device_triggers_and_finalize_against_the_reference():
    """All golden frames as ONE batch (Batch.from_data_list layout): the device chain's trigger bits are exactly "a node with flow > 3" /
    "a cluster with more than four members" per frame, and postprocess.finalize returns the reference's final predictions and partition."""
    import torch

    from gnn_cca_amd.postprocess import finalize, prune_and_cluster
    eis, probs, node_ptr, edge_ptr, want_pred, want_ids, want_trig = [], [], [0], [0], [], [], []
    for name in NAMES:
        n, ei, _, p, g = case(name)
        eis.append(ei + node_ptr[-1])
        probs.append(p)
        fo, fi = po.flows(ei, g("pred_pruned"), n)
        want_trig.append((1 if (fo > 3).any() or (fi > 3).any() else 0) | (2 if np.bincount(g("id_pruned")).max() > 4 else 0))
        want_pred.append(g("pred_final"))
        want_ids.append(g("id_final") + 1000 * len(want_ids))
        node_ptr.append(node_ptr[-1] + n)
        edge_ptr.append(edge_ptr[-1] + ei.shape[1])
    ei_all, probs_all = np.concatenate(eis, axis=1), np.concatenate(probs)
    dev = torch.device("cuda", 0)
    ei_d, probs_d = torch.from_numpy(ei_all).to(dev), torch.from_numpy(probs_all).to(dev)
    preds_d = (probs_d >= 0.5).long()
    post = prune_and_cluster(ei_d, preds_d, node_ptr[-1], node_ptr, edge_ptr)
    assert post["triggers"].cpu().numpy().tolist() == want_trig
    assert sum(1 for t in want_trig if t) >= 10 and want_trig.count(0) >= 1
    fin = finalize(ei_d, probs_d, post["pruned"], post["labels"], post["n_clusters"], post["triggers"], node_ptr, edge_ptr)
    assert fin["frames_finalized"] == [g for g, t in enumerate(want_trig) if t]
    assert np.array_equal(fin["predictions"].cpu().numpy(), np.concatenate(want_pred))
    want_all = np.concatenate(want_ids)
    assert po.same_partition(fin["labels"].cpu().numpy(), want_all)
    assert int(fin["n_clusters"].item()) == len(set(want_all.tolist()))
    # the switches: no SPLITTING = the stored `rounded` results
    fin_r = finalize(ei_d, probs_d, post["pruned"], post["labels"], post["n_clusters"], post["triggers"], node_ptr, edge_ptr, splitting=False)
    assert np.array_equal(fin_r["predictions"].cpu().numpy(), np.concatenate([case(nm)[4]("pred_rounded") for nm in NAMES]))
    # the single-graph form (no frame ranges): one frame = the first golden case
    n0, ei0, _, p0, g0 = case(NAMES[0])
    e0, p0d = torch.from_numpy(ei0).to(dev), torch.from_numpy(p0).to(dev)
    post0 = prune_and_cluster(e0, (p0d >= 0.5).long(), n0)
    fin0 = finalize(e0, p0d, post0["pruned"], post0["labels"], post0["n_clusters"], post0["triggers"])
    assert np.array_equal(fin0["predictions"].cpu().numpy(), g0("pred_final")) and po.same_partition(fin0["labels"].cpu().numpy(), g0("id_final"))

"""Row N2 of SURVEY.md 8f: the step right after the MPN on the GPU.

`threshold` = inference.py:286-291 (sigmoid, >= 0.5); `prune_and_cluster` = utils.remove_edges_single_direction
(libs/utils.py:387-404) + the flow counts of utils.compute_rounding (libs/utils.py:54-59) + the clusters
utils.compute_SCC_and_Clusters (libs/utils.py:295-317) returns for the pruned edge set -- without the networkx /
Python-list round trips through the host -- and, per frame, the two TRIGGER bits of the reference's bridge-based heuristics
(a node with flow > 3: utils.compute_rounding has work to do, libs/utils.py:58-62; a cluster with more than four members:
utils.disjoint_big_clusters has, libs/utils.py:321-322).  `finalize` runs those heuristics, in the order of the shipped
configuration (config_inference.yaml:6-8 ROUNDING / PRUNING / SPLITTING = True; inference.py:306-345), for the frames that
raised a bit: native host code (csrc/post_host.cpp; SURVEY.md 8f keeps the bridge searches on the CPU), one frame at a time
as the reference does.  A frame with no bit set already has its final partition.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat


from .graph_build import _on, _raw_stream as _stream   # the lean device guard / raw stream handle (host time matters here: ~6 launches)


def threshold(logits):
    """logits: Tensor[E] or [E,1] on the GPU -> (probs float32 [E], predictions int64 [E])."""
    if not logits.is_cuda:
        raise RuntimeError("gnn_cca_amd.postprocess runs on MI355X only (no CPU fallback)")
    x = logits.reshape(-1)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    probs = torch.empty_like(x)
    preds = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    with _on(x.device):
        st = nat.lib().gnncca_post_threshold(x.data_ptr(), x.shape[0], probs.data_ptr(), preds.data_ptr(), _stream(x.device))
    if st:
        nat.check(st, "gnncca_post_threshold")
    return probs, preds


def prune_and_cluster(edge_index, predictions, n_nodes, node_ptr=None, edge_ptr=None):
    """-> dict(pruned int64 [E], flow_out / flow_in int32 [N], labels int32 [N] (smallest node id of the component),
    n_clusters int32 [1] on the device).

    `node_ptr` / `edge_ptr` (optional, both or neither): the frame ranges of a batch laid out like
    Batch.from_data_list -- sequences of G + 1 ints or int32 device tensors (GraphBatch.node_ptr / .edge_ptr, or the
    device copies build_graph_batch keeps).  With them every frame is clustered by its own workgroup."""
    if not (edge_index.is_cuda and predictions.is_cuda):
        raise RuntimeError("gnn_cca_amd.postprocess runs on MI355X only (no CPU fallback)")
    dev = edge_index.device
    ei = edge_index if edge_index.dtype == torch.int64 and edge_index.is_contiguous() else edge_index.long().contiguous()
    pred = predictions.reshape(-1)
    if pred.dtype != torch.int64 or not pred.is_contiguous():
        pred = pred.long().contiguous()
    e = ei.shape[1]
    lib = nat.lib()
    ws = torch.empty(lib.gnncca_post_workspace_bytes(n_nodes, e) + 256, dtype=torch.uint8, device=dev)
    if (node_ptr is None) != (edge_ptr is None):
        raise ValueError("node_ptr and edge_ptr go together")
    n_trig = (len(node_ptr) if not torch.is_tensor(node_ptr) else node_ptr.numel()) - 1 if node_ptr is not None else 1
    # flow_out | flow_in | n_clusters | cluster sizes (scratch) | triggers: zeroed by ONE memset
    counters = torch.empty(3 * n_nodes + 1 + max(n_trig, 1), dtype=torch.int32, device=dev)
    out = {"pruned": torch.empty(e, dtype=torch.int64, device=dev),
           "flow_out": counters[:n_nodes], "flow_in": counters[n_nodes:2 * n_nodes],
           "labels": torch.empty(n_nodes, dtype=torch.int32, device=dev),
           "n_clusters": counters[2 * n_nodes:2 * n_nodes + 1],
           "triggers": counters[3 * n_nodes + 1:3 * n_nodes + 1 + max(n_trig, 1)]}
    n_frames, np_dev, ep_dev = 0, None, None
    if node_ptr is not None:
        def as_dev(v):
            if torch.is_tensor(v):
                if v.dtype == torch.int32 and v.device == dev and v.is_contiguous():
                    return v
                return v.to(device=dev, dtype=torch.int32).contiguous()
            return torch.tensor(list(v), dtype=torch.int32).to(dev)
        np_dev, ep_dev = as_dev(node_ptr), as_dev(edge_ptr)
        n_frames = np_dev.numel() - 1
        if ep_dev.numel() != n_frames + 1 or n_frames < 1:
            raise ValueError("node_ptr / edge_ptr must both have G + 1 entries")
    with _on(dev):
        cp = counters.data_ptr()
        st = lib.gnncca_post_prune_cluster_frames_ex(ei.data_ptr(), pred.data_ptr(), n_nodes, e,
                                                     np_dev.data_ptr() if np_dev is not None else None,
                                                     ep_dev.data_ptr() if ep_dev is not None else None, n_frames,
                                                     ws.data_ptr(), ws.numel(), out["pruned"].data_ptr(),
                                                     cp, cp + 4 * n_nodes, out["labels"].data_ptr(), cp + 8 * n_nodes,
                                                     cp + 8 * n_nodes + 4, cp + 12 * n_nodes + 4, _stream(dev))
    if st:
        nat.check(st, "gnncca_post_prune_cluster")
    out["_workspace"] = (ws, np_dev, ep_dev)
    return out


def finalize(edge_index, probs, pruned, labels, n_clusters, triggers, node_ptr=None, edge_ptr=None, rounding=True, pruning=True,
             splitting=True):
    """The reference's FINAL predictions and identity clusters under the three switches of config_inference.yaml:6-8 (inference.py:306-345:
    PRUNING -> ROUNDING -> PRUNING -> SPLITTING), from what `threshold` + `prune_and_cluster` left on the device.

    `triggers` [G] (prune_and_cluster's) says which frames the heuristics can change; this call SYNCHRONISES to read it.  Frames with no
    bit set keep the device chain's result -- it is final for them.  For the others the frame's edges, probabilities and pruned predictions
    are copied to the host, `gnncca_post_finalize_frame_host` (csrc/post_host.cpp: utils.compute_rounding, remove_edges_single_direction,
    disjoint_big_clusters, compute_SCC_and_Clusters with the reference's artefacts) runs frame by frame, and the patched tensors go back.
    `node_ptr` / `edge_ptr`: HOST sequences of G + 1 ints (GraphBatch.node_ptr / .edge_ptr); None = the whole graph is one frame.

    -> dict(predictions int64 [E], labels int32 [N], n_clusters int32 [1] on the device, frames_finalized: list of frame ids,
            triggers: host int32 [G]).  `pruning=False` is refused: the device chain has already pruned."""
    if not pruning:
        raise ValueError("finalize works on the pruned predictions of prune_and_cluster (PRUNING = True, as shipped)")
    trig = triggers.cpu().numpy()          # the one synchronisation of this stage
    want = (nat.POST_TRIGGER_ROUNDING if rounding else 0) | (nat.POST_TRIGGER_SPLITTING if splitting else 0)
    todo = np.nonzero(trig & want)[0].tolist()
    if not todo:
        return {"predictions": pruned, "labels": labels, "n_clusters": n_clusters, "frames_finalized": [], "triggers": trig}
    n, e = labels.shape[0], pruned.shape[0]
    if node_ptr is None:
        node_ptr, edge_ptr = [0, n], [0, e]
    node_ptr, edge_ptr = [int(v) for v in node_ptr], [int(v) for v in edge_ptr]
    ei = edge_index.cpu().numpy()
    pr = np.ascontiguousarray(probs.reshape(-1).cpu().numpy(), dtype=np.float32)
    pred = pruned.cpu().numpy().copy()
    lab = labels.cpu().numpy().copy()
    src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    total = int(n_clusters.cpu().numpy()[0])
    switches = (nat.POST_ROUNDING if rounding else 0) | nat.POST_PRUNING | (nat.POST_SPLITTING if splitting else 0)
    lib = nat.lib()
    np_h, ep_h = np.asarray(node_ptr, dtype=np.int32), np.asarray(edge_ptr, dtype=np.int32)
    listed, k_new = np.asarray(todo, dtype=np.int32), np.zeros(len(todo), dtype=np.int32)
    roots = lab == np.arange(n, dtype=lab.dtype)                       # the device chain's labels: a component's smallest node id
    before = sum(int(roots[np_h[g]:np_h[g + 1]].sum()) for g in todo)
    st = lib.gnncca_post_finalize_frames_host(src.ctypes.data, dst.ctypes.data, np_h.ctypes.data, ep_h.ctypes.data, listed.ctypes.data, len(todo),
                                              pr.ctypes.data, pred.ctypes.data, switches, lab.ctypes.data, k_new.ctypes.data, 0)
    if st:
        nat.check(st, "gnncca_post_finalize_frames_host")
    total += int(k_new.sum()) - before
    dev = pruned.device
    return {"predictions": torch.from_numpy(pred).to(dev), "labels": torch.from_numpy(lab).to(dev),
            "n_clusters": torch.tensor([total], dtype=torch.int32, device=dev), "frames_finalized": todo, "triggers": trig}

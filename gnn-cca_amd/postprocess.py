"""Row N2 of SURVEY.md 8f: the step right after the MPN on the GPU.

`threshold` = inference.py:286-291 (sigmoid, >= 0.5); `prune_and_cluster` = utils.remove_edges_single_direction
(libs/utils.py:387-404) + the flow counts of utils.compute_rounding (libs/utils.py:54-59) + the clusters
utils.compute_SCC_and_Clusters (libs/utils.py:295-317) returns for the pruned edge set -- without the networkx /
Python-list round trips through the host.  The bridge-based rounding and splitting heuristics are not reproduced here.
"""
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
    counters = torch.empty(2 * n_nodes + 1, dtype=torch.int32, device=dev)   # flow_out | flow_in | n_clusters: zeroed by ONE memset
    out = {"pruned": torch.empty(e, dtype=torch.int64, device=dev),
           "flow_out": counters[:n_nodes], "flow_in": counters[n_nodes:2 * n_nodes],
           "labels": torch.empty(n_nodes, dtype=torch.int32, device=dev),
           "n_clusters": counters[2 * n_nodes:]}
    if (node_ptr is None) != (edge_ptr is None):
        raise ValueError("node_ptr and edge_ptr go together")
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
        st = lib.gnncca_post_prune_cluster_frames(ei.data_ptr(), pred.data_ptr(), n_nodes, e,
                                                  np_dev.data_ptr() if np_dev is not None else None,
                                                  ep_dev.data_ptr() if ep_dev is not None else None, n_frames,
                                                  ws.data_ptr(), ws.numel(), out["pruned"].data_ptr(),
                                                  out["flow_out"].data_ptr(), out["flow_in"].data_ptr(),
                                                  out["labels"].data_ptr(), out["n_clusters"].data_ptr(), _stream(dev))
    if st:
        nat.check(st, "gnncca_post_prune_cluster")
    out["_workspace"] = (ws, np_dev, ep_dev)
    return out

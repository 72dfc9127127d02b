"""Row N1 of SURVEY.md 8f: the step right before the MPN -- building the cross-camera graph and its edge attributes
(`inference.py:189-279`, duplicated at `train.py:257-361` and `train.py:616-692`) -- on the GPU.

The reference does this per frame with Python list comprehensions, sklearn on the host and several GPU<->CPU round
trips (its real end-to-end bottleneck).  Here the host only derives the edge ENUMERATION from the camera ids (O(N)
numpy, `plan_frames`); one HIP kernel (`gnncca_build_edges`) then writes `edge_index`, `edge_attr` and `edge_labels`
for the whole batch of frames, and `gnncca_normalize_columns` does the `F.normalize(..., dim=0)` of the embeddings.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as nat
from .sharding import GraphBatch

MODE_FULL, MODE_ONLY_APPEARANCE, MODE_ONLY_DIST = 0, 1, 2


@dataclass
class FramePlan:
    src_order: np.ndarray   # [N] int32  node ids in the order the reference emits their out-edges
    edge_ptr: np.ndarray    # [N+1] int32 first edge of each source position
    graph_ptr: np.ndarray   # [G+1] int32 node range of each frame graph
    graph_of: np.ndarray    # [N] int32
    n_edges: int


def plan_frames(id_cam, graph_sizes):
    """Edge enumeration of inference.py:207-212: per graph, cameras in np.unique order; within a camera its nodes in
    ascending id; each connects to every node of the other cameras in ascending id."""
    id_cam = np.asarray(id_cam)
    sizes = np.asarray(graph_sizes, dtype=np.int64)
    graph_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(graph_ptr[-1])
    if len(id_cam) != n:
        raise ValueError("id_cam length does not match graph_sizes")
    graph_of = np.repeat(np.arange(len(sizes), dtype=np.int32), sizes)
    # one stable sort of (graph, camera) keys for the whole batch: graph-major, np.unique's camera order inside a graph,
    # node id ascending inside a camera
    _, cam_rank = np.unique(id_cam, return_inverse=True)
    n_cam = int(cam_rank.max()) + 1 if n else 1
    key = graph_of.astype(np.int64) * n_cam + cam_rank
    src_order = np.argsort(key, kind="stable").astype(np.int32)
    same_cam = np.bincount(key, minlength=len(sizes) * n_cam)
    deg = (sizes[graph_of] - same_cam[key])[src_order]
    edge_ptr = np.concatenate([[0], np.cumsum(deg)])
    if edge_ptr[-1] >= 2 ** 31 - 64:
        raise NotImplementedError("more than 2^31 edges in one batch")
    return FramePlan(src_order, edge_ptr.astype(np.int32), graph_ptr, graph_of, int(edge_ptr[-1]))


def normalize_columns(x):
    """F.normalize(x, p=2, dim=0) (inference.py:189-190) as a HIP kernel pair; returns a new tensor."""
    if not x.is_cuda:
        raise RuntimeError("gnn_cca_amd.graph_build runs on MI355X only (no CPU fallback)")
    x = x.float().contiguous()
    out = torch.empty_like(x)
    scratch = torch.empty(((x.shape[0] + 63) // 64 + 1) * x.shape[1], dtype=torch.float32, device=x.device)  # 64-row chunk sums + norms
    with torch.cuda.device(x.device):
        st = nat.lib().gnncca_normalize_columns(x.data_ptr(), x.shape[0], x.shape[1], scratch.data_ptr(), out.data_ptr(),
                                                torch.cuda.current_stream(x.device).cuda_stream)
    nat.check(st, "gnncca_normalize_columns")
    return out


def build_graph_batch(xw, yw, ids, id_cam, graph_sizes, max_dist, node_embeds, reid_embeds, only_appearance=False,
                      only_dist=False, normalize=True):
    """One call per batch of frames.  Host inputs (numpy, one entry per detection, frames concatenated): xw, yw, ids,
    id_cam; graph_sizes / max_dist per frame.  Device inputs: node_embeds [N, D], reid_embeds [N, R].
    Returns a GraphBatch (x, edge_index, edge_attr) with .edge_labels and .y, laid out exactly like the reference's
    `Batch.from_data_list(batch)` (inference.py:279)."""
    if not (node_embeds.is_cuda and reid_embeds.is_cuda):
        raise RuntimeError("gnn_cca_amd.graph_build runs on MI355X only (no CPU fallback)")
    dev = reid_embeds.device
    plan = plan_frames(id_cam, graph_sizes)
    n, e = len(plan.src_order), plan.n_edges
    if reid_embeds.shape[0] != n or node_embeds.shape[0] != n:
        raise RuntimeError("embeddings and detections disagree on the number of nodes")
    if normalize:
        reid_embeds = normalize_columns(reid_embeds)
        node_embeds = normalize_columns(node_embeds)
    else:
        reid_embeds = reid_embeds.float().contiguous()
    mode = MODE_ONLY_APPEARANCE if only_appearance else (MODE_ONLY_DIST if only_dist else MODE_FULL)
    n_attr = 4 if mode == MODE_FULL else 2
    _, dense_ids = np.unique(np.asarray(ids), return_inverse=True)
    # ONE host->device transfer with every per-node / per-graph array: 8-byte fields first, then the int32 ones
    g = len(plan.graph_ptr) - 1
    ids64 = np.ascontiguousarray(ids, dtype=np.int64)
    f64 = np.concatenate([np.asarray(xw, np.float64), np.asarray(yw, np.float64), np.asarray(max_dist, np.float64)])
    edge_ptr_g = plan.edge_ptr[plan.graph_ptr]  # edges are emitted graph by graph: graph g owns [edge_ptr_g[g], edge_ptr_g[g+1])
    i32 = np.concatenate([dense_ids.astype(np.int32), np.asarray(id_cam).astype(np.int32), plan.graph_of, plan.graph_ptr,
                          plan.src_order, plan.edge_ptr, edge_ptr_g.astype(np.int32)])
    host = np.concatenate([f64.view(np.uint8), ids64.view(np.uint8), i32.view(np.uint8)])
    staged = torch.from_numpy(host).to(dev)
    fr = nat.Frames()
    p0 = staged.data_ptr()
    fr.xw, fr.yw, fr.max_dist = p0, p0 + 8 * n, p0 + 16 * n
    y_off = 8 * (2 * n + g)
    base, o = p0 + y_off + 8 * n, 0
    for name, cnt in (("person_id", n), ("cam", n), ("graph_of", n), ("graph_ptr", g + 1), ("src_order", n), ("edge_ptr", n + 1)):
        setattr(fr, name, base + 4 * o)
        o += cnt
    edge_index = torch.empty((2, e), dtype=torch.int64, device=dev)
    edge_attr = torch.empty((e, n_attr), dtype=torch.float32, device=dev)
    edge_labels = torch.empty(e, dtype=torch.float32, device=dev)
    if e > 0:
        with torch.cuda.device(dev):
            st = nat.lib().gnncca_build_edges(C.byref(fr), reid_embeds.data_ptr(), reid_embeds.shape[1], n, e, mode,
                                              edge_index.data_ptr(), edge_attr.data_ptr(), edge_labels.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream)
        nat.check(st, "gnncca_build_edges")
    # per-graph edge ranges: edges are emitted graph by graph, so graph g owns edge_ptr[graph_ptr[g]] .. edge_ptr[graph_ptr[g+1]]
    batch = GraphBatch(node_embeds, edge_index, edge_attr, edge_ptr_g.tolist(), plan.graph_ptr.tolist())
    # device copies of the frame ranges (int32 [G + 1]) for the per-frame post-processing (postprocess.prune_and_cluster)
    i32_dev = staged[y_off + 8 * n:].view(torch.int32)
    batch.node_ptr_dev = i32_dev[3 * n:3 * n + g + 1]
    batch.edge_ptr_dev = i32_dev[5 * n + g + 2:5 * n + 2 * g + 3]
    batch.edge_labels = edge_labels
    batch.y = staged[y_off:y_off + 8 * n].view(torch.int64)
    batch.reid_embeds = reid_embeds
    batch._keepalive = staged
    return batch

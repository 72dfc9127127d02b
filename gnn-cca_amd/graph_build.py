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


def _raw_stream(dev):
    return torch._C._cuda_getCurrentRawStream(dev.index)


_stream_objs = {}


def _current_stream(dev):
    """torch.cuda.current_stream(dev) through a cache keyed by the raw handle (the call itself costs ~10 us of host time)."""
    raw = _raw_stream(dev)
    hit = _stream_objs.get(dev.index)
    if hit is None or hit[0] != raw:
        hit = _stream_objs[dev.index] = (raw, torch.cuda.current_stream(dev))
    return hit[1]


class _on:
    """`with torch.cuda.device(dev)` only when `dev` is not the current device already (the context manager costs ~5 us of host
    time per use, more than a launch)."""

    def __init__(self, dev):
        self._ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


FUSED_NORMALIZE_MAX_ROWS = 4096   # gnncca_normalize_columns2: one launch for up to two matrices of a batch of frames


def _f32c(x):
    return x if x.dtype == torch.float32 and x.is_contiguous() else x.float().contiguous()


def normalize_columns(x, other=None):
    """F.normalize(x, p=2, dim=0) (inference.py:189-190) on the GPU; returns a new tensor -- or, given a second matrix with the same
    number of rows (`other`: the reid and the node embeddings of a batch), the pair, normalised in one launch when the batch has at
    most 4096 rows (same bits as the three-kernel form, which takes any size)."""
    if not x.is_cuda or (other is not None and not other.is_cuda):
        raise RuntimeError("gnn_cca_amd.graph_build runs on MI355X only (no CPU fallback)")
    x = _f32c(x)
    out = torch.empty_like(x)
    if other is not None:
        other = _f32c(other)
        if other.shape[0] != x.shape[0]:
            raise ValueError("normalize_columns(x, other): both matrices must have the same number of rows")
        out2 = torch.empty_like(other)
    lib = nat.lib()
    with _on(x.device):
        if x.shape[0] <= FUSED_NORMALIZE_MAX_ROWS and x.dim() == 2:
            st = lib.gnncca_normalize_columns2(x.data_ptr(), x.shape[1], out.data_ptr(), other.data_ptr() if other is not None else None,
                                               other.shape[1] if other is not None else 0, out2.data_ptr() if other is not None else None,
                                               x.shape[0], _raw_stream(x.device))
        else:
            st = 0
            for a, o in ((x, out),) + (((other, out2),) if other is not None else ()):
                scratch = torch.empty(((a.shape[0] + 63) // 64 + 1) * a.shape[1], dtype=torch.float32, device=a.device)  # 64-row chunk sums + norms
                st = st or lib.gnncca_normalize_columns(a.data_ptr(), a.shape[0], a.shape[1], scratch.data_ptr(), o.data_ptr(), _raw_stream(a.device))
    if st:
        nat.check(st, "gnncca_normalize_columns")
    return out if other is None else (out, out2)


class _Staging:
    """Ring of pinned host buffers for the per-batch staging image (gnncca_plan_frames writes it, ONE non-blocking copy uploads
    it): the host never waits for the GPU, so the graph of the next batch of frames is planned while this one's kernels run.
    A slot is reused only after the copy that read it has completed (its event)."""
    SLOTS = 8

    def __init__(self):
        self._bufs, self._events, self._next = [None] * self.SLOTS, [None] * self.SLOTS, 0

    def take(self, nbytes):
        i = self._next
        self._next = (i + 1) % self.SLOTS
        if self._events[i] is not None:
            self._events[i].synchronize()
        buf = self._bufs[i]
        if buf is None or buf.numel() < nbytes:
            buf = self._bufs[i] = torch.empty(max(2 * nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)
        if self._events[i] is None:
            self._events[i] = torch.cuda.Event()
        return buf, self._events[i]


_staging = {}


def _as(a, dtype):
    a = np.asarray(a)
    if a.dtype != dtype or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=dtype)
    return a


def build_graph_batch(xw, yw, ids, id_cam, graph_sizes, max_dist, node_embeds, reid_embeds, only_appearance=False,
                      only_dist=False, normalize=True):
    """One call per batch of frames.  Host inputs (numpy, one entry per detection, frames concatenated): xw, yw, ids,
    id_cam; graph_sizes / max_dist per frame.  Device inputs: node_embeds [N, D], reid_embeds [N, R].
    Returns a GraphBatch (x, edge_index, edge_attr) with .edge_labels and .y, laid out exactly like the reference's
    `Batch.from_data_list(batch)` (inference.py:279).
    Host work: one native call that enumerates the edges (gnncca_plan_frames, include/gnncca_mpn.h) into a pinned staging buffer,
    one non-blocking upload, four launches; the call never synchronises."""
    if not (node_embeds.is_cuda and reid_embeds.is_cuda):
        raise RuntimeError("gnn_cca_amd.graph_build runs on MI355X only (no CPU fallback)")
    dev = reid_embeds.device
    lib = nat.lib()
    xw, yw, md = _as(xw, np.float64), _as(yw, np.float64), _as(max_dist, np.float64)
    ids64, cam64, sizes = _as(ids, np.int64), _as(id_cam, np.int64), _as(graph_sizes, np.int64)
    n, g = len(cam64), len(sizes)
    if not (len(xw) == len(yw) == len(ids64) == n) or len(md) != g:
        raise ValueError("per-detection / per-frame arrays disagree on their lengths")
    if reid_embeds.shape[0] != n or node_embeds.shape[0] != n:
        raise RuntimeError("embeddings and detections disagree on the number of nodes")
    nbytes = lib.gnncca_plan_frames_bytes(n, g)
    ring = _staging.get(dev.index)
    if ring is None:
        ring = _staging[dev.index] = _Staging()
    pinned, event = ring.take(nbytes)
    e = lib.gnncca_plan_frames(xw.ctypes.data, yw.ctypes.data, ids64.ctypes.data, cam64.ctypes.data, n, sizes.ctypes.data, md.ctypes.data, g,
                               pinned.data_ptr(), nbytes)
    if e < 0:
        if -e == nat.ERR_INVALID_ARG:
            raise ValueError("id_cam length does not match graph_sizes")
        nat.check(int(-e), "gnncca_plan_frames")
    with _on(dev):
        staged = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        staged.copy_(pinned[:nbytes], non_blocking=True)
        event.record(_current_stream(dev))
        if normalize:
            reid_embeds, node_embeds = normalize_columns(reid_embeds, node_embeds)
        elif reid_embeds.dtype != torch.float32 or not reid_embeds.is_contiguous():
            reid_embeds = reid_embeds.float().contiguous()
        mode = MODE_ONLY_APPEARANCE if only_appearance else (MODE_ONLY_DIST if only_dist else MODE_FULL)
        n_attr = 4 if mode == MODE_FULL else 2
        fr = nat.Frames()
        p0 = staged.data_ptr()
        fr.xw, fr.yw, fr.max_dist = p0, p0 + 8 * n, p0 + 16 * n
        y_off = 8 * (2 * n + g)
        i32_off = y_off + 8 * n
        base = p0 + i32_off
        fr.person_id, fr.cam, fr.graph_of = base, base + 4 * n, base + 8 * n
        fr.graph_ptr, fr.src_order, fr.edge_ptr = base + 12 * n, base + 4 * (3 * n + g + 1), base + 4 * (4 * n + g + 1)
        edge_index = torch.empty((2, e), dtype=torch.int64, device=dev)
        edge_attr = torch.empty((e, n_attr), dtype=torch.float32, device=dev)
        edge_labels = torch.empty(e, dtype=torch.float32, device=dev)
        if e > 0:
            st = lib.gnncca_build_edges(C.byref(fr), reid_embeds.data_ptr(), reid_embeds.shape[1], n, e, mode,
                                        edge_index.data_ptr(), edge_attr.data_ptr(), edge_labels.data_ptr(), _raw_stream(dev))
            if st:
                nat.check(st, "gnncca_build_edges")
    # per-graph ranges (host copies): graph g owns the nodes graph_ptr[g] .. graph_ptr[g+1] and, edges being emitted graph by graph,
    # the edges edge_ptr_g[g] .. edge_ptr_g[g+1]
    host_i32 = pinned[i32_off:nbytes].numpy().view(np.int32)
    node_ptr = host_i32[3 * n:3 * n + g + 1].tolist()
    edge_ptr = host_i32[5 * n + g + 2:5 * n + 2 * g + 3].tolist()
    batch = GraphBatch(node_embeds, edge_index, edge_attr, edge_ptr, node_ptr)
    # device copies of the frame ranges (int32 [G + 1]) for the per-frame post-processing (postprocess.prune_and_cluster)
    i32_dev = staged[i32_off:].view(torch.int32)
    batch.node_ptr_dev = i32_dev[3 * n:3 * n + g + 1]
    batch.edge_ptr_dev = i32_dev[5 * n + g + 2:5 * n + 2 * g + 3]
    batch.edge_labels = edge_labels
    batch.y = staged[y_off:y_off + 8 * n].view(torch.int64)
    batch.reid_embeds = reid_embeds
    batch._keepalive = staged
    return batch

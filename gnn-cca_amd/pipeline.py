"""A batch of frames from detections to identity clusters in ONE native call (rows N1 + the path + N2 of SURVEY.md 8f).

The per-batch body of the reference's inference loop (inference.py:189-345: normalise the embeddings, build the cross-camera graph,
`mpn_model(data_batch)`, sigmoid / threshold, prune single-direction edges, flow counts, identity clusters) runs here as the SAME
launches `graph_build.build_graph_batch`, `MOTMPNet.forward`, `postprocess.threshold` and `postprocess.prune_and_cluster` make -- the
results are bit for bit theirs (tests/test_gpu_pipeline.py) -- but issued from one call of the C ABI (`gnncca_frames_forward`).  At this
size (a batch of 64 Terrace frames: 1229 detections, 21 930 edges, 15 launches of 3-15 us) the loop is bound by host time per launch,
and the Python between five calls is a third of it: 0.143 -> 0.11 ms per batch.

    pipe = FramePipeline(model)                                     # model: gnn_cca_amd.MOTMPNet on the GPU, eval mode
    r = pipe(xw, yw, ids, id_cam, graph_sizes, max_dist, node_embeds, reid_embeds)
    r.labels, r.n_clusters, r.pruned, r.preds, r.probs, r.outputs['classified_edges'], r.batch (x, edge_index, edge_attr, ...)
    r.triggers                                                      # [G] int32 on the device: frames the host heuristics can change
    f = r.final()                                                   # the reference's FINAL predictions / ID_pred partition under
    f['predictions'], f['labels'], f['n_clusters']                  # ROUNDING / PRUNING / SPLITTING (config_inference.yaml:6-8)

`r.final_async()` (round 6) hands the batch to a persistent pool of host threads WITHOUT synchronising -- one D2H copy of the batch's trigger
words / edges / probabilities / pruned predictions / labels enqueued behind the chain, the flagged frames finalized by the
pool while the caller enqueues the next batch -- and returns a `PendingFinal`; `.result()` waits for that batch only:

    pending = [pipe(*batch_k).final_async() for ...]                # batch k's host pass overlaps batch k + 1's GPU chain
    f = pending[k].result()                                         # host arrays: f['predictions'], f['labels'], f['n_clusters'], ...

`r.final()` is `final_async().result()` with the results uploaded again (device tensors, as `postprocess.finalize` returns them); frames that raised none keep the device chain's result
(it is final for them), the others go through the reference's rounding / splitting heuristics on the host (csrc/post_host.cpp), frame by
frame as the reference's batch-size-1 validation loop does.  The constructor's `rounding` / `pruning` / `splitting` mirror CONFIG's keys.

Batches of more than 4096 detections, train mode and forward hooks take the step-by-step path (same results).  No CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from .graph_build import MODE_FULL, MODE_ONLY_APPEARANCE, MODE_ONLY_DIST, _as, _current_stream, _on, _raw_stream, _Staging, _staging, build_graph_batch
from .postprocess import finalize, prune_and_cluster, threshold
from .sharding import GraphBatch

MAX_NODES = 4096


class PendingFinal:
    """A batch on its way through the host heuristics (gnncca_post_pool_*).  `result()` blocks until THIS batch is final -- it never
    synchronises the device -- and returns host arrays: predictions int64 [E], labels int32 [N] (a cluster's smallest node id), n_clusters
    int, frames_finalized (the frames that went through the heuristics), triggers int32 [G].  The arrays are copies unless `copy=False`
    (views of this object's pinned buffer: valid while the object lives)."""

    def __init__(self, pipe, ticket, host, views, keep):
        self._pipe, self._ticket, self._host, self._views, self._keep = pipe, ticket, host, views, keep
        self._res = None

    def done(self):
        return self._res is not None

    def result(self, copy=True):
        if self._res is None:
            host, start, off, n, e, g = self._views
            frames, count = np.empty(max(g, 1), dtype=np.int32), C.c_int32(0)
            times = np.zeros(4, dtype=np.float64)
            st = nat.lib().gnncca_post_pool_wait_timed(self._pipe._pool_handle(), self._ticket, frames.ctypes.data, C.byref(count), times.ctypes.data)
            self.times_us = times      # submit -> picked up -> event complete -> last frame final -> collected
            self._keep = None      # the device buffers may go: everything is on the host now
            self._pipe._outstanding -= 1
            if st:
                nat.check(st, "gnncca_post_pool_wait")
            hn = host.numpy()

            def view(key, dtype, cnt):
                o = off[key] - start
                return hn[o:o + cnt * np.dtype(dtype).itemsize].view(dtype)

            self._res = {"predictions": view("pruned", np.int64, e), "labels": view("labels", np.int32, n),
                         "n_clusters": int(view("n_clusters", np.int32, 1)[0]),
                         "frames_finalized": frames[:count.value].tolist(), "triggers": view("triggers", np.int32, g)}
        if not copy:
            return self._res
        return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self._res.items()}

    def __del__(self):
        try:
            pool = self._pipe._pool
            if self._res is None and self._ticket is not None and pool:   # never collected: wait, or the pool would write into freed memory
                nat.lib().gnncca_post_pool_wait(pool, self._ticket, None, None)
                self._pipe._outstanding -= 1
            self._pipe._give_back(self._host)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class FrameResult:
    """Outputs of one batch; tensors are views of ONE device buffer owned by this object."""
    __slots__ = ("batch", "outputs", "probs", "preds", "pruned", "flow_out", "flow_in", "labels", "n_clusters", "triggers", "_switches", "_final", "_keep",
                 "_pipe", "_d2h", "_pending")

    def final_async(self):
        """Hand this batch to the pipeline's pool of host threads (no synchronisation) -> PendingFinal.  Results that did not come from the
        one-call path (more than 4096 detections, hooks, train mode) are finalized on the spot."""
        if self._pending is None:
            self._pending = self._pipe._submit_final(self) if self._d2h is not None else _Finished(self)
        return self._pending

    def final(self):
        """The reference's final predictions and identity clusters for this batch (inference.py:306-345 under the pipeline's ROUNDING /
        PRUNING / SPLITTING switches) as DEVICE tensors, computed once: `final_async().result()` uploaded again (or, for results of the
        step-by-step path, `postprocess.finalize`).  Waits for this batch."""
        if self._final is None:
            if self._d2h is None:
                b = self.batch
                r, p, s = self._switches
                self._final = finalize(b.edge_index, self.probs, self.pruned, self.labels, self.n_clusters, self.triggers, b.node_ptr, b.edge_ptr,
                                       rounding=r, pruning=p, splitting=s)
            else:
                res = self.final_async().result(copy=False)
                dev = self.pruned.device
                if not res["frames_finalized"]:      # nothing changed: the device chain's tensors ARE the final result
                    self._final = {"predictions": self.pruned, "labels": self.labels, "n_clusters": self.n_clusters,
                                   "frames_finalized": [], "triggers": res["triggers"]}
                else:   # upload from the PINNED buffer itself (torch views of it: truly asynchronous copies on the current stream, which is also
                    #        the stream the next batch's D2H into a recycled buffer would be ordered behind; pageable uploads: 0.59-0.64 -> 0.48-0.49 ms
                    #        per batch, same box)
                    host, start, off, n, e, _g = self._pending._views
                    pv = host[off["pruned"] - start:off["pruned"] - start + 8 * e].view(torch.int64)
                    lv = host[off["labels"] - start:off["labels"] - start + 4 * n].view(torch.int32)
                    kv = host[off["n_clusters"] - start:off["n_clusters"] - start + 4].view(torch.int32)
                    self._final = {"predictions": pv.to(dev, non_blocking=True), "labels": lv.to(dev, non_blocking=True),
                                   "n_clusters": kv.to(dev, non_blocking=True),
                                   "frames_finalized": res["frames_finalized"], "triggers": res["triggers"]}
        return self._final


class _Finished:
    """PendingFinal of a result that was finalized synchronously (the step-by-step path)."""

    def __init__(self, r):
        f = r.final()
        self._res = {"predictions": f["predictions"].cpu().numpy(), "labels": f["labels"].cpu().numpy(), "n_clusters": int(f["n_clusters"].item()),
                     "frames_finalized": f["frames_finalized"], "triggers": f["triggers"]}

    def done(self):
        return True

    def result(self, copy=True):
        return dict(self._res)


class FramePipeline:
    def __init__(self, model, only_appearance=False, only_dist=False, normalize=True, rounding=True, pruning=True, splitting=True):
        self.model = model
        if not pruning:
            raise ValueError("FramePipeline prunes on the device (PRUNING = True, as config_inference.yaml:7 ships it)")
        self.switches = (bool(rounding), bool(pruning), bool(splitting))
        self.mode = MODE_ONLY_APPEARANCE if only_appearance else (MODE_ONLY_DIST if only_dist else MODE_FULL)
        self.normalize = bool(normalize)
        self._post_ws = {}   # (stream, bytes) -> workspace tensor
        self._shape = None   # (n, e) of the cached workspace sizes
        self._sizes = (0, 0)
        self._pool = None    # gnncca_post_pool (created on the first final_async)
        self._pinned = {}    # bytes (power of two) -> free pinned host buffers of that size
        self.host_threads = 0   # 0: the library's default (hardware threads - 2, at most 16)
        self._outstanding = 0   # batches submitted to the pool and not yet collected

    def _pool_handle(self):
        if self._pool is None:
            self._pool = nat.lib().gnncca_post_pool_create(int(self.host_threads))
            if not self._pool:
                raise nat.NativeError("gnncca_post_pool_create failed")
        return self._pool

    def close(self):
        """Stops the pool's threads.  Every PendingFinal must have been collected (or dropped) first: a batch still on its way is refused."""
        if self._outstanding > 0:
            raise RuntimeError(f"{self._outstanding} batch(es) are still with the host pool: collect their PendingFinal.result() before close()")
        if self._pool is not None:
            nat.lib().gnncca_post_pool_destroy(self._pool)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _take_pinned(self, nbytes):
        size = 1 << max(12, int(nbytes - 1).bit_length())
        free = self._pinned.setdefault(size, [])
        return free.pop() if free else torch.empty(size, dtype=torch.uint8, pin_memory=True)

    def _give_back(self, host):
        if host is not None and self._pinned is not None:
            self._pinned.setdefault(host.numel(), []).append(host)

    def _submit_final(self, r):
        """One D2H copy of [probs | edge_index | pruned | counters | labels] (contiguous in the batch's arena) behind the chain on its stream, and
        the job behind that copy: nothing here waits for the device."""
        arena, start, end, off, n, e, g = r._d2h
        dev = arena.device
        nbytes = end - start
        host = self._take_pinned(nbytes)
        hp = host.data_ptr()
        views = (host, start, off, n, e, g)      # numpy views are made when the result is collected
        node_ptr, edge_ptr = np.asarray(r.batch.node_ptr, dtype=np.int32), np.asarray(r.batch.edge_ptr, dtype=np.int32)
        b = nat.PostBatch()
        b.src, b.dst = hp + off["edge_index"] - start, hp + off["edge_index"] - start + 8 * e
        b.node_ptr, b.edge_ptr, b.n_frames = node_ptr.ctypes.data, edge_ptr.ctypes.data, g
        rr, _, ss = r._switches
        b.switches = (nat.POST_ROUNDING if rr else 0) | nat.POST_PRUNING | (nat.POST_SPLITTING if ss else 0)
        b.triggers, b.probs = hp + off["triggers"] - start, hp + off["probs"] - start
        b.predictions, b.labels, b.n_clusters = hp + off["pruned"] - start, hp + off["labels"] - start, hp + off["n_clusters"] - start
        with _on(dev):
            ticket = nat.lib().gnncca_post_pool_submit_copy(self._pool_handle(), C.byref(b), arena.data_ptr() + start, hp, nbytes, dev.index,
                                                            _raw_stream(dev))
        if ticket < 0:
            self._give_back(host)
            nat.check(int(-ticket), "gnncca_post_pool_submit_copy")
        self._outstanding += 1
        return PendingFinal(self, ticket, host, views, (r._keep, node_ptr, edge_ptr))

    def _slow(self, xw, yw, ids, id_cam, sizes, max_dist, node, reid):
        b = build_graph_batch(xw, yw, ids, id_cam, sizes, max_dist, node, reid, only_appearance=self.mode == MODE_ONLY_APPEARANCE,
                              only_dist=self.mode == MODE_ONLY_DIST, normalize=self.normalize)
        with torch.no_grad():
            out = self.model(b)
        r = FrameResult()
        r.batch, r.outputs = b, out
        r.probs, r.preds = threshold(out["classified_edges"][-1])
        post = prune_and_cluster(b.edge_index, r.preds, b.x.shape[0], b.node_ptr_dev, b.edge_ptr_dev)
        r.pruned, r.flow_out, r.flow_in, r.labels, r.n_clusters = post["pruned"], post["flow_out"], post["flow_in"], post["labels"], post["n_clusters"]
        r.triggers, r._switches, r._final = post["triggers"], self.switches, None
        r._keep = post
        r._pipe, r._d2h, r._pending = self, None, None
        return r

    def __call__(self, xw, yw, ids, id_cam, graph_sizes, max_dist, node_embeds, reid_embeds):
        m = self.model
        if not (node_embeds.is_cuda and reid_embeds.is_cuda):
            raise RuntimeError("gnn_cca_amd.pipeline runs on MI355X only (no CPU fallback)")
        n = int(node_embeds.shape[0])
        if m.training or n > MAX_NODES or n == 0 or m._containers_hooked():
            return self._slow(xw, yw, ids, id_cam, graph_sizes, max_dist, node_embeds, reid_embeds)
        dev = reid_embeds.device
        lib = nat.lib()
        xw, yw, md = _as(xw, np.float64), _as(yw, np.float64), _as(max_dist, np.float64)
        ids64, cam64, sizes = _as(ids, np.int64), _as(id_cam, np.int64), _as(graph_sizes, np.int64)
        g = len(sizes)
        if not (len(xw) == len(yw) == len(ids64) == len(cam64) == n) or len(md) != g or reid_embeds.shape[0] != n:
            raise ValueError("per-detection / per-frame arrays disagree on their lengths")
        d = m.native_dims()
        if node_embeds.dim() != 2 or node_embeds.shape[1] != d.node_in or node_embeds.dtype != torch.float32 or reid_embeds.dtype != torch.float32:
            raise RuntimeError(f"expected float32 embeddings [N, {d.node_in}] / [N, R], got {tuple(node_embeds.shape)} {node_embeds.dtype}, "
                               f"{tuple(reid_embeds.shape)} {reid_embeds.dtype}")
        node_embeds = node_embeds if node_embeds.is_contiguous() else node_embeds.contiguous()
        reid_embeds = reid_embeds if reid_embeds.is_contiguous() else reid_embeds.contiguous()
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
        if e == 0:   # no cross-camera pair in the whole batch: nothing to launch on the path (the step-by-step functions return the empty containers)
            return self._slow(xw, yw, ids64, cam64, sizes, md, node_embeds, reid_embeds)
        n_attr = 4 if self.mode == MODE_FULL else 2
        if n_attr != d.edge_in:
            raise RuntimeError(f"the model takes {d.edge_in} edge attributes, this pipeline's mode builds {n_attr}")
        r_dim, d_in = int(reid_embeds.shape[1]), int(d.node_in)
        with _on(dev):
            blob = m._packed_weights(dev)
            hot = m._hot
            if hot.n_out < 0:
                hot.n_out = lib.gnncca_num_outputs(C.byref(d))
            n_out = hot.n_out
            # three buffers: fp32 (normalised embeddings, edge attributes, labels, logits, probabilities), int64, int32; every region
            # starts on a 256-byte boundary (the kernels use 16-byte accesses where the pointers allow them)
            def up(v):
                return (v + 63) // 64 * 64
            o_node, o_reid = 0, up(n * d_in) if self.normalize else 0
            o_attr = o_reid + up(n * r_dim) if self.normalize else 0
            o_lab = o_attr + up(e * n_attr)
            o_log = o_lab + up(e)
            o_prob = o_log + up(n_out * e)
            o_ei, o_prun = 0, up(2 * e)       # (the thresholded predictions live in front of the f32 block: final_async's D2H region is
            #                                      [probs | edge_index | pruned | counters | labels] and does not carry them)
            o_labels = up(3 * n + 1 + g)      # flow_out | flow_in | n_clusters | cluster sizes (scratch) | triggers [G]
            if self._shape != (n, e):
                self._shape = (n, e)
                self._sizes = (lib.gnncca_workspace_bytes(C.byref(d), n, e) if e > 0 else 0, lib.gnncca_post_workspace_bytes(n, e) + 256)
            ws_bytes, post_bytes = self._sizes
            if e > 0 and ws_bytes == 0:
                nat.check(lib.gnncca_supported(C.byref(d)), "MOTMPNet configuration")
            ws = m._scratch(max(ws_bytes, 256), dev)
            # ONE allocation per batch (round 5; five until then, 2-3 us of host time each): [staging image | predictions | f32 | i64 | i32 | post workspace],
            # every region on a 256-byte boundary
            def up256(v):
                return (v + 255) // 256 * 256
            b_pred = up256(nbytes)
            b_f32 = b_pred + up256(8 * e)
            b_i64 = b_f32 + up256(4 * (o_prob + e))
            b_i32 = b_i64 + up256(8 * (o_prun + e))
            b_post = b_i32 + up256(4 * (o_labels + n))
            arena = torch.empty(b_post + post_bytes, dtype=torch.uint8, device=dev)
            staged = arena[:nbytes]
            staged.copy_(pinned[:nbytes], non_blocking=True)
            event.record(_current_stream(dev))
            f32 = arena[b_f32:b_f32 + 4 * (o_prob + e)].view(torch.float32)
            i64 = arena[b_i64:b_i64 + 8 * (o_prun + e)].view(torch.int64)
            preds = arena[b_pred:b_pred + 8 * e].view(torch.int64)
            i32 = arena[b_i32:b_i32 + 4 * (o_labels + n)].view(torch.int32)
            post_ws = arena[b_post:b_post + post_bytes]
            io = nat.FramesIO()
            io.staged_dev, io.n_nodes, io.n_frames, io.n_edges = staged.data_ptr(), n, g, e
            io.node_embeds, io.reid_embeds, io.reid_dim, io.mode, io.normalize = node_embeds.data_ptr(), reid_embeds.data_ptr(), r_dim, self.mode, int(self.normalize)
            fp, ip, cp = f32.data_ptr(), i64.data_ptr(), i32.data_ptr()
            if self.normalize:
                io.node_norm, io.reid_norm = fp + 4 * o_node, fp + 4 * o_reid
            io.edge_attr, io.edge_labels = fp + 4 * o_attr, fp + 4 * o_lab
            io.logits, io.probs = fp + 4 * o_log, fp + 4 * o_prob
            io.edge_index, io.predictions, io.pruned = ip + 8 * o_ei, preds.data_ptr(), ip + 8 * o_prun
            io.counters, io.labels, io.counters_len = cp, cp + 4 * o_labels, o_labels
            st = lib.gnncca_frames_forward(C.byref(d), blob.data_ptr(), C.byref(io), ws.data_ptr(), ws.numel(), post_ws.data_ptr(), post_ws.numel(),
                                           m._options(), _raw_stream(dev))
        if st:
            nat.check(st, "gnncca_frames_forward")
        # views (the reference's containers): x, edge_index, edge_attr, logits as [E, 1] per classified step
        x = f32[o_node:o_node + n * d_in].view(n, d_in) if self.normalize else node_embeds
        reid_n = f32[o_reid:o_reid + n * r_dim].view(n, r_dim) if self.normalize else reid_embeds
        i32_off = 8 * (3 * n + g)
        host_i32 = pinned[i32_off:nbytes].numpy().view(np.int32)
        batch = GraphBatch(x, i64[o_ei:o_ei + 2 * e].view(2, e), f32[o_attr:o_attr + e * n_attr].view(e, n_attr),
                           host_i32[5 * n + g + 2:5 * n + 2 * g + 3].tolist(), host_i32[3 * n:3 * n + g + 1].tolist())
        i32_dev = staged[i32_off:].view(torch.int32)
        batch.node_ptr_dev = i32_dev[3 * n:3 * n + g + 1]
        batch.edge_ptr_dev = i32_dev[5 * n + g + 2:5 * n + 2 * g + 3]
        batch.edge_labels = f32[o_lab:o_lab + e]
        batch.y = staged[8 * (2 * n + g):8 * (3 * n + g)].view(torch.int64)
        batch.reid_embeds = reid_n
        r = FrameResult()
        r.batch = batch
        r.outputs = {"classified_edges": list(f32[o_log:o_log + n_out * e].view(n_out, e, 1).unbind(0))}
        r.probs = f32[o_prob:o_prob + e]
        r.preds, r.pruned = preds, i64[o_prun:o_prun + e]
        r.flow_out, r.flow_in, r.n_clusters, r.labels = i32[:n], i32[n:2 * n], i32[2 * n:2 * n + 1], i32[o_labels:o_labels + n]
        r.triggers, r._switches, r._final = i32[3 * n + 1:3 * n + 1 + g], self.switches, None
        r._keep = (arena, ws, blob)
        # final_async: ONE contiguous D2H region of the arena, [probs .. labels] (byte offsets inside the arena)
        off = {"probs": b_f32 + 4 * o_prob, "edge_index": b_i64 + 8 * o_ei, "pruned": b_i64 + 8 * o_prun, "n_clusters": b_i32 + 4 * (2 * n),
               "triggers": b_i32 + 4 * (3 * n + 1), "labels": b_i32 + 4 * o_labels}
        r._pipe, r._pending = self, None
        r._d2h = (arena, off["probs"], b_i32 + 4 * (o_labels + n), off, n, e, g)
        return r

"""HIP-graph replay of the eval forward for the reference's per-frame inference loop.

The caller of the hot path at inference time is a loop over frames, `inference.py:173-283`: build the frame's graph, then
`outputs = mpn_model(data_batch)` (inference.py:283) -- one forward per frame.  On this path a forward of a frame-sized graph is
six kernel launches of 4-7 us each, i.e. it is bound by launching, not by the kernels: an eager call costs
the host ~30 us, about what the GPU needs.  `MOTMPNet.forward` in eval mode allocates nothing but its outputs, never
synchronises and enqueues only on the current stream, so the whole call can be captured ONCE per input shape into a HIP graph and
replayed: `GraphedForward` does that, in three forms.

    gf = GraphedForward(model)                       # model: gnn_cca_amd.MOTMPNet, on the GPU, eval mode
    for data in frames:
        out = gf(data)                               # same dict as model(data); valid until the next call of this shape
        preds = out['classified_edges'][-1].view(-1) # inference.py:286

  * `gf(data)`: capture per input shape (after `warmup` eager calls of that shape), STATIC input buffers: the frame is copied
    into the buffers the graph was captured on (three device-to-device copies on the current stream) and the graph is replayed.
    A caller that produces its frames IN those buffers (`gf.static_inputs(data)`, e.g. as the output tensors of the graph-build
    step) skips the copies: `gf(static)` sees its own tensors and replays at once.
  * `gf.block(frames)`: K frames captured BACK TO BACK in one HIP graph (K whole forwards, every launch of each, each with its own
    outputs) -- the K-deep form: one graph launch per K frames, the launch-bound loop runs without the host in it.
    `blk.replay()` -> list of K output dicts; `blk.inputs[i]` are the static buffers of frame i.  `gf.block(frames, chains=S, depth=D)`
    cuts them into graphs of D frames replayed round robin on S streams (S frames in flight, one workspace per stream).
  * `streams=S`: S independent forwards in flight, each stream replaying its own graph on its own workspace (the module keeps a
    workspace per stream; the packed weights are shared and read-only).  `gf.submit(data)` returns a `Pending` whose
    `.result()` makes the CURRENT stream wait for that forward only; a producer that writes its frames into the slots' own static
    inputs (`gf.slot_inputs`) drives them with `gf.replay_slot(i)` / `gf.join()` -- one graph launch of host work per frame.

Replays are bitwise the eager forward (same kernels, same launch parameters, same order: tests/test_gpu_inference_graph.py).
No CPU fallback, no caching of results: every replay runs every kernel of every forward it holds.
"""
import torch

from .mpn import _raw_stream


class _Frame:
    """Duck-typed `data` of MOTMPNet.forward (models/mpn.py:266 reads .x, .edge_index, .edge_attr only)."""
    __slots__ = ("x", "edge_index", "edge_attr")

    def __init__(self, x, edge_index, edge_attr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr


def _key(data):
    return (tuple(data.x.shape), tuple(data.edge_index.shape), tuple(data.edge_attr.shape), data.x.device.index)


def _check(model, data):
    if model.training:
        raise RuntimeError("GraphedForward replays the eval forward: call model.eval() (training steps have GraphedTrainStep)")
    if not (data.x.is_cuda and data.edge_index.is_cuda and data.edge_attr.is_cuda):
        raise RuntimeError("gnn_cca_amd.inference runs on MI355X only: move the module and `data` to the GPU (there is no CPU fallback)")
    model._check_inputs(data.x, data.edge_index, data.edge_attr)   # the reference's dtype contract (RuntimeError)


def _clone(data):
    return _Frame(data.x.detach().clone().contiguous(), data.edge_index.clone().contiguous(), data.edge_attr.detach().clone().contiguous())


def _copy_into(static, data):
    """The frame into the captured buffers (nothing when the caller already works in them)."""
    if data.x.data_ptr() != static.x.data_ptr():
        static.x.copy_(data.x, non_blocking=True)
    if data.edge_index.data_ptr() != static.edge_index.data_ptr():
        static.edge_index.copy_(data.edge_index, non_blocking=True)
    if data.edge_attr.data_ptr() != static.edge_attr.data_ptr():
        static.edge_attr.copy_(data.edge_attr, non_blocking=True)


class GraphedBlock:
    """K forwards captured in ONE HIP graph back to back, or (chains = S) in S graphs replayed on S streams (GraphedForward.block)."""

    def __init__(self, graphs, streams, inputs, outputs, stamp, workspaces, owner=None, blob=None):
        self._graphs, self._streams, self.inputs, self.outputs, self._stamp, self._workspaces = graphs, streams, inputs, outputs, stamp, workspaces
        self._owner, self._blob = owner, blob   # the packed weights the graphs address stay alive with them
        self._uniq_streams = [st for i, st in enumerate(streams) if st is not None and st not in streams[:i]]

    def __len__(self):
        return len(self.inputs)

    def replay(self, frames=None):
        """Run the K forwards: one graph launch on the current stream -- or, with S chains, one launch per chain on its own stream,
        forked from and joined back into the current stream by events (no host synchronisation).  `frames`: K new frames to copy into
        the static inputs first (same shapes as at capture; omit when the producer wrote them there).  Returns the K output dicts --
        static tensors, valid until the next replay."""
        if self._owner is not None and self._owner._stamp(self.inputs[0].x.device) != self._stamp:
            # (a weight update repacks into the SAME buffer and is seen by the replay; what invalidates a block is another packed buffer --
            # the generic family's host packer, .to() -- or another option / step count: other kernels, other addresses)
            raise RuntimeError("the module's packed weights or options changed since this block was captured: ask GraphedForward.block() again")
        if frames is not None:
            if len(frames) != len(self.inputs):
                raise ValueError(f"this block holds {len(self.inputs)} frames, got {len(frames)}")
            for s, f in zip(self.inputs, frames):
                if _key(f) != _key(s):
                    raise ValueError("frame shapes differ from the captured block's")
                _copy_into(s, f)
        if len(self._graphs) == 1:
            self._graphs[0].replay()
            return self.outputs
        cur = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        for st in self._uniq_streams:
            st.wait_event(fork)
        try:   # groups of <= `depth` frames, round robin over the S streams (set_stream, not the context manager: a third of its host cost,
            #    and a short block is bound by how fast its first launches go out)
            for g, st in zip(self._graphs, self._streams):
                torch.cuda.set_stream(st)
                g.replay()
        finally:
            torch.cuda.set_stream(cur)
        for st in self._uniq_streams:
            cur.wait_stream(st)
        return self.outputs


class Pending:
    """A forward in flight on one of GraphedForward's streams."""

    def __init__(self, outputs, event):
        self._outputs, self._event = outputs, event

    def result(self):
        """The outputs, ordered after that forward on the CURRENT stream (no host synchronisation)."""
        torch.cuda.current_stream().wait_event(self._event)
        return self._outputs


class GraphedForward:
    def __init__(self, model, warmup=2, max_graphs=64, streams=1):
        self.model = model
        self.warmup, self.max_graphs = int(warmup), int(max_graphs)
        self._seen = {}      # shape key -> eager calls so far
        self._static = {}    # shape key -> static frame handed out by static_inputs() before the capture
        self._graphs = {}    # (shape key, slot) -> (graph, static frame, outputs, stamp, workspace kept alive)
        self._blocks = {}    # tuple of shape keys -> GraphedBlock
        # the sequential forms (gf(data), gf.block(...)) capture on ONE private stream, i.e. on one grow-only workspace of the module:
        # their replays are meant for one stream at a time (use submit() for forwards in flight together)
        self._cap_stream = None
        self._chain_streams = []   # streams of block(..., chains=S): one graph and one workspace each
        self._streams = [torch.cuda.Stream() for _ in range(int(streams))] if int(streams) > 1 else []
        self._next_slot = 0
        self._slot_last, self._slot_out = {}, {}   # slot -> (graph, stream) / outputs of the shape last used on it (replay_slot)

    # -- what a captured graph depends on besides the input shapes: the packed weight blob's address and the module's options ------
    def _stamp(self, device):
        m = self.model
        blob = m._packed_weights(device)   # repacks (eagerly, on the current stream, into the same buffer) if the parameters changed
        return (blob.data_ptr(), m._options(), int(m.num_enc_steps), int(m.num_class_steps))

    def _drop_stale(self, stamp):
        stale = [k for k, v in self._graphs.items() if v[3] != stamp]
        dropped = {id(self._graphs[k][0]) for k in stale}
        for k in stale:
            del self._graphs[k]
        # a slot must not keep replaying a dropped graph: it addresses the workspace and the packed blob that went with it
        for slot in [s for s, (g, _st) in self._slot_last.items() if id(g) in dropped]:
            del self._slot_last[slot]
            self._slot_out.pop(slot, None)
        for k in [k for k, b in self._blocks.items() if b._stamp != stamp]:
            del self._blocks[k]

    def _capture(self, frames, stream=None):
        """Capture model(frame) for every frame of the list, back to back, on `stream` (default: this object's capture stream).
        Returns (graph, outputs, the module workspace the graph addresses -- the caller keeps it alive with the graph)."""
        m = self.model
        if stream is None:
            if self._cap_stream is None:
                # ONE stream serves the sequential forms and chain 0: this device maps streams onto four hardware queues in the order
                # they are first used, the caller's stream holds one, and a chain stream that lands on an occupied queue does not
                # overlap (measured: chains = 3 after a capture on a stream of its own 26 us per forward, with the shared stream 16)
                if not self._chain_streams:
                    self._chain_streams.append(torch.cuda.Stream())
                self._cap_stream = self._chain_streams[0]
            stream = self._cap_stream
        with torch.no_grad():
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):      # the stream's workspace reaches its final size before the capture begins
                big = max(frames, key=lambda f: (f.edge_index.shape[1], f.x.shape[0]))
                m(big)
                for f in frames:
                    if f is not big and _key(f) != _key(big):
                        m(f)
            torch.cuda.current_stream().wait_stream(stream)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                outs = [m(f) for f in frames]
            ws = m._hot.workspace
        return graph, outs, ws

    # -- one frame per call ------------------------------------------------------------------------------------------------------------
    def static_inputs(self, data):
        """The static input buffers of this frame shape (allocated on first use): write the next frame INTO them and pass the returned
        object to `gf(...)` to replay without copies."""
        _check(self.model, data)
        key = (_key(data), 0)
        entry = self._graphs.get(key)
        if entry is not None:
            return entry[1]
        pre = self._static.get(key)
        if pre is None:
            pre = self._static[key] = _clone(data)
        return pre

    def __call__(self, data):
        _check(self.model, data)
        dev = data.x.device
        stamp = self._stamp(dev)
        key = (_key(data), 0)
        entry = self._graphs.get(key)
        if entry is not None and entry[3] == stamp:
            graph, static, outs = entry[:3]
            _copy_into(static, data)
            graph.replay()
            return outs
        if data.x.shape[0] == 0 or data.edge_index.shape[1] == 0:
            with torch.no_grad():
                return self.model(data)     # nothing to launch: the eager call returns empty logits
        self._drop_stale(stamp)
        n = self._seen.get(key, 0)
        if n < self.warmup or len(self._graphs) >= self.max_graphs:
            self._seen[key] = n + 1
            with torch.no_grad():
                return self.model(data)
        static = self._static.pop(key, None) or _clone(data)
        _copy_into(static, data)
        graph, outs, ws = self._capture([static])
        graph.replay()   # capture records, it does not execute: this replay is this call's forward
        self._graphs[key] = (graph, static, outs[0], stamp, ws, self.model._packed[1])
        return outs[0]

    # -- K frames per graph launch -----------------------------------------------------------------------------------------------------
    def block(self, frames, adopt_inputs=False, chains=1, depth=4):
        """Capture (once per sequence of shapes) the forwards of `frames` in one HIP graph.  `adopt_inputs=True`: the given
        tensors ARE the static buffers (they stay resident in HBM and the producer overwrites them in place); otherwise they are
        cloned.  The same object may appear several times (one forward per appearance, each with its own outputs).
        `chains=S` > 1: S independent frames in flight instead of one chain in frame order (a frame-sized forward is six dependent
        launches that leave most of the chip idle): the frames are cut into groups of `depth` consecutive frames, every group is one
        HIP graph, and the groups are replayed round robin on S streams (a workspace per stream).  `depth` trades host launches
        (one per group: ~15 us each, which is what a frame-sized forward takes on the GPU) against how soon the streams overlap:
        one graph per stream (depth = K / S) barely overlaps -- the host-side launch of a 400-kernel graph takes a third of its run
        -- and so do parallel branches inside ONE graph (25.8 vs 27.6 us per forward at three branches)."""
        frames = list(frames)
        if not frames:
            raise ValueError("empty block")
        for f in frames:
            _check(self.model, f)
            if f.x.shape[0] == 0 or f.edge_index.shape[1] == 0:
                raise ValueError("a block cannot hold empty frames")
        dev = frames[0].x.device
        stamp = self._stamp(dev)
        self._drop_stale(stamp)
        key = (tuple(_key(f) for f in frames), tuple(f.x.data_ptr() for f in frames) if adopt_inputs else None, int(chains), int(depth))
        blk = self._blocks.get(key)
        if blk is None:
            clones = {}
            inputs = []
            for f in frames:
                if adopt_inputs:
                    inputs.append(_Frame(f.x, f.edge_index, f.edge_attr))
                else:
                    if id(f) not in clones:
                        clones[id(f)] = _clone(f)
                    inputs.append(clones[id(f)])
            chains = max(1, min(int(chains), len(inputs)))
            if chains == 1:
                graph, outs, ws = self._capture(inputs)
                blk = GraphedBlock([graph], [None], inputs, outs, stamp, [ws], self, self.model._packed[1])
            else:   # groups of `depth` consecutive frames, one graph each, group g on stream g % S (its own workspace of the module)
                while len(self._chain_streams) < chains:
                    self._chain_streams.append(torch.cuda.Stream())
                depth = max(1, int(depth))
                # groups of <= `depth` consecutive frames, their number a multiple of S and their sizes within one frame of each other, so
                # that every stream carries the same number of frames (20 frames, depth 2, S = 3: 7 / 7 / 6 frames per stream; plain groups
                # of two gave 8 / 6 / 6).  Measured at K = 20 (tools/archive/chains_sweep.py 20): 19.7-20.7 us per forward for every depth from 1 to
                # 7 -- a short block is bound by its fixed costs (fork, three staggered first launches, join: ~70 us), not by the cut)
                k = len(inputs)
                n_groups = min(k, chains * -(-k // (chains * depth)))
                sizes = [k // n_groups + (1 if g < k % n_groups else 0) for g in range(n_groups)]
                graphs, streams, wss, outs = [], [], [], [None] * k
                lo = 0
                for gi, size in enumerate(sizes):
                    idx = list(range(lo, lo + size))
                    lo += size
                    st = self._chain_streams[gi % chains]
                    g, o, ws = self._capture([inputs[i] for i in idx], stream=st)
                    graphs.append(g)
                    streams.append(st)
                    wss.append(ws)
                    for i, oi in zip(idx, o):
                        outs[i] = oi
                blk = GraphedBlock(graphs, streams, inputs, outs, stamp, wss, self, self.model._packed[1])
            self._blocks[key] = blk
        return blk

    # -- S forwards in flight ------------------------------------------------------------------------------------------------------------
    def slot_inputs(self, data, slot):
        """The static input buffers of stream `slot` (0 ... S-1) for this frame shape, allocated and captured on first use: write a
        frame INTO them and `submit` the returned object to replay on that stream without copies."""
        if not self._streams:
            raise RuntimeError("GraphedForward(model, streams=S) with S > 1 is needed for slot_inputs()")
        _check(self.model, data)
        return self._slot_entry(data, int(slot) % len(self._streams), self._stamp(data.x.device))[1]

    def _slot_entry(self, data, slot, stamp):
        key = (_key(data), 1 + slot)
        entry = self._graphs.get(key)
        if entry is None or entry[3] != stamp:
            self._drop_stale(stamp)
            static = _clone(data)
            graph, outs, ws = self._capture([static], stream=self._streams[slot])
            entry = self._graphs[key] = (graph, static, outs[0], stamp, ws, self.model._packed[1])
        self._slot_last[slot], self._slot_out[slot] = (entry[0], self._streams[slot]), entry[2]
        return entry

    def replay_slot(self, slot):
        """The lean form of `submit` for a producer that keeps every slot's static inputs filled (`slot_inputs`) and orders itself:
        replay slot `slot`'s graph of the shape last handed out / submitted on it, on the slot's stream -- no checks, no copies, no
        fork, no event (one graph launch of host work).  Order the results with `join()`."""
        last = self._slot_last.get(slot)
        if last is None:
            raise RuntimeError(f"replay_slot({slot}): no live graph on this slot (never submitted, or dropped when the weights / options "
                               f"changed): hand the frame to submit() / slot_inputs() again")
        graph, st = last
        with torch.cuda.stream(st):
            graph.replay()

    def slot_outputs(self, slot):
        """The (static) outputs of the graph `replay_slot(slot)` replays."""
        return self._slot_out[slot]

    def join(self):
        """The CURRENT stream waits for everything enqueued on the S streams so far (no host synchronisation)."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            cur.wait_stream(st)

    def submit(self, data, after_current=True):
        """Enqueue this frame's forward on one of the S streams and return at once: the stream whose static inputs `data` IS
        (`slot_inputs`: no copies), else the next one round robin (the frame is copied into that stream's static inputs).  Each
        stream has its own graph, static inputs and workspace per frame shape.  `after_current=True` orders the copy and the replay
        after the work already on the CURRENT stream (the producer of `data`); a producer that writes the slot's inputs on the slot's
        own stream, or synchronises by itself, passes False and saves the fork (an event record + a stream wait per call).
        A slot's outputs are reused by its next frame: take `.result()` (or copy) before submitting to the same slot again."""
        if not self._streams:
            raise RuntimeError("GraphedForward(model, streams=S) with S > 1 is needed for submit()")
        _check(self.model, data)
        dev = data.x.device
        stamp = self._stamp(dev)
        slot = None
        for i in range(len(self._streams)):   # is `data` one of the slots' own static frames?
            e = self._graphs.get((_key(data), 1 + i))
            if e is not None and e[1] is data:
                slot = i
                break
        if slot is None:
            slot = self._next_slot
            self._next_slot = (slot + 1) % len(self._streams)
        st = self._streams[slot]
        graph, static, outs = self._slot_entry(data, slot, stamp)[:3]
        if after_current:
            st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            _copy_into(static, data)
            graph.replay()
            ev = torch.cuda.Event()
            ev.record(st)
        return Pending(outs, ev)


class PaddedForward:
    """ONE HIP graph for every frame of the per-frame loop (inference.py:173-283), whatever its (N, E).

    `GraphedForward` keeps a graph per input shape; a camera sequence has hundreds of shapes (the 4816 Terrace frames:
    tests/golden/terrace_topology.npz).  Here every frame is padded to ONE canonical shape instead -- `n_max` real nodes plus
    `n_dummy` dummy nodes behind them, `e_max` edges -- by one kernel launch (`gnncca_pad_frame`, include/gnncca_mpn.h): rows beyond
    N are zero, the missing edges are self loops with zero attributes on the dummy nodes.  The dummy nodes are a component of
    their own, and models/mpn.py has no term that crosses components (what Batch.from_data_list relies on, inference.py:279): the
    logits of the frame's own edges are those of the frame alone (to rounding: the encoder's reduction order depends on the row
    count).  Per frame the host enqueues the pad kernel and one graph launch; the outputs are VIEWS of the first E rows of the
    graph's static logits, valid until the slot's next frame.

        pf = PaddedForward(model, n_max=40, e_max=1200)          # streams=S: S frames in flight, round robin
        for data in frames:
            preds = pf(data)['classified_edges'][-1].view(-1)     # inference.py:283-286

    A frame that does not fit (N > n_max or E > e_max) runs the eager forward; so does an empty one."""

    def __init__(self, model, n_max, e_max, n_dummy=8, streams=1):
        from . import _native as nat
        self._nat = nat
        self.model, self.n_max, self.e_max, self.n_dummy = model, int(n_max), int(e_max), max(1, int(n_dummy))
        self._gf = GraphedForward(model, warmup=0, streams=streams)
        self._s = max(1, int(streams))
        self._slots, self._stamp0 = None, None
        self._next = 0
        self.padded, self.eager = 0, 0   # frames served by the graph / by the eager forward (did not fit)

    def _template(self, data):
        n, e = self.n_max + self.n_dummy, self.e_max
        x = torch.zeros((n, data.x.shape[1]), dtype=data.x.dtype, device=data.x.device)
        ea = torch.zeros((e, data.edge_attr.shape[1]), dtype=data.edge_attr.dtype, device=data.x.device)
        ei = torch.full((2, e), self.n_max, dtype=torch.int64, device=data.x.device)   # a valid graph for the capture: loops on one dummy node
        return _Frame(x, ei, ea)

    def _make_slots(self, data):
        t = self._template(data)
        gf = self._gf
        if self._s == 1:
            # capture now (GraphedForward(warmup=0) captures on the first call of a shape); the template's buffers become the static inputs
            static = gf.static_inputs(t)
            gf(static)
            graph, _, outs = gf._graphs[(_key(static), 0)][:3]
            self._slots = [(static, graph, outs, None)]
        else:
            self._slots = []
            for i in range(self._s):
                static = gf.slot_inputs(t, i)
                graph, st = gf._slot_last[i]
                self._slots.append((static, graph, gf._slot_out[i], st))
        self._node_in, self._edge_in = int(data.x.shape[1]), int(data.edge_attr.shape[1])
        self._pad_fn = self._nat.lib().gnncca_pad_frame
        torch.cuda.synchronize()

    def __call__(self, data):
        """Returns the dict of model(data) -- for streams > 1 a `Pending` (`.result()` orders the current stream behind that frame)."""
        _check(self.model, data)
        n, e = data.x.shape[0], data.edge_index.shape[1]
        if n == 0 or e == 0 or n > self.n_max or e > self.e_max:
            self.eager += 1
            with torch.no_grad():
                out = self.model(data)
            if self._s > 1:
                ev = torch.cuda.Event()
                ev.record()
                return Pending(out, ev)
            return out
        stamp = self._gf._stamp(data.x.device)
        if self._slots is None or stamp != self._stamp0:   # first frame, or the graphs address other weights / kernels now: capture again
            self._gf._drop_stale(stamp)
            self._make_slots(data)
            self._stamp0 = stamp
        if data.x.dim() != 2 or data.edge_attr.dim() != 2 or data.x.shape[1] != self._node_in or data.edge_attr.shape[1] != self._edge_in \
                or data.edge_index.dim() != 2 or data.edge_index.shape[0] != 2 or data.edge_attr.shape[0] != e:
            raise RuntimeError(f"shape mismatch: x {tuple(data.x.shape)}, edge_index {tuple(data.edge_index.shape)}, edge_attr "
                               f"{tuple(data.edge_attr.shape)} for node_in={self._node_in}, edge_in={self._edge_in}")
        self.padded += 1
        x, ei, ea = data.x, data.edge_index, data.edge_attr
        if not (x.is_contiguous() and ei.is_contiguous() and ea.is_contiguous()):
            x, ei, ea = x.contiguous(), ei.contiguous(), ea.contiguous()
        i = self._next
        static, graph, outs, st = self._slots[i]
        if st is None:
            self._pad(x, ei, ea, static, n, e, _raw_stream(x.device))
            graph.replay()
            return {"classified_edges": [t[:e] for t in outs["classified_edges"]]}
        self._next = i + 1 if i + 1 < self._s else 0
        st.wait_stream(torch.cuda.current_stream())     # the producer of `data`
        self._pad(x, ei, ea, static, n, e, st.cuda_stream)
        with torch.cuda.stream(st):
            graph.replay()
        ev = torch.cuda.Event()
        ev.record(st)
        for t in (x, ei, ea):
            t.record_stream(st)
        return Pending({"classified_edges": [t[:e] for t in outs["classified_edges"]]}, ev)

    def _pad(self, x, ei, ea, static, n, e, raw_stream):
        status = self._pad_fn(x.data_ptr(), n, ei.data_ptr(), ea.data_ptr(), e, static.x.data_ptr(), self.n_max, self.n_dummy,
                              static.edge_index.data_ptr(), static.edge_attr.data_ptr(), self.e_max, self._node_in, self._edge_in, raw_stream)
        if status:
            self._nat.check(status, "gnncca_pad_frame")

    def join(self):
        self._gf.join()

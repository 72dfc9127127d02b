"""Multi-GPU layer of the path: graph-level sharding, one process per GPU.

Frame graphs are independent (disjoint union in Batch.from_data_list, inference.py:279; no cross-graph edges) and the
weights are shared and read-only in eval mode, so the path shards by GRAPH with no data-path collective:

  * rank r of W owns graphs [lo, hi) of the batch (`shard_range`), builds its own disjoint union with local node
    numbering (`union_graphs`) and runs the ordinary single-GPU forward on it;
  * the only collective is a one-time broadcast of the shared MLP weights (the state_dict as one flat 1.07 MB buffer)
    from rank 0 over RCCL/xGMI (`broadcast_weights`; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests); every
    rank then packs its own kernel blob from them (byte-identical everywhere).

The reference has no distributed code at all (single GPU, main_training.py:7).
"""
import torch


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items over `world` ranks; the first n_items % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GraphBatch:
    """Duck-typed `data` for MOTMPNet.forward: x, edge_index, edge_attr (+ per-graph edge offsets)."""

    def __init__(self, x, edge_index, edge_attr, edge_ptr, node_ptr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr
        self.edge_ptr, self.node_ptr = edge_ptr, node_ptr


def union_graphs(graphs):
    """Disjoint union of graphs given as (x [n, D], edge_index [2, e] with LOCAL node ids, edge_attr [e, A]):
    node ids are shifted by the running node count, edges are concatenated in graph order -- the same layout
    torch_geometric's Batch.from_data_list produces (inference.py:279), so `row` stays sorted if it was."""
    xs, eis, eas, edge_ptr, node_ptr = [], [], [], [0], [0]
    for x, ei, ea in graphs:
        eis.append(ei + node_ptr[-1])
        xs.append(x)
        eas.append(ea)
        node_ptr.append(node_ptr[-1] + x.shape[0])
        edge_ptr.append(edge_ptr[-1] + ei.shape[1])
    if not xs:
        raise ValueError("empty shard")
    return GraphBatch(torch.cat(xs), torch.cat(eis, dim=1), torch.cat(eas), edge_ptr, node_ptr)


class PerGraphLogits:
    """Sequence over the graphs of a GraphBatch; item g = list over classified steps of Tensor[e_g, 1] (views of the
    batch's logits, made when asked for: slicing 512 graphs x 3 steps eagerly costs more host time than the whole GPU
    forward of that batch)."""

    def __init__(self, outputs, batch):
        self._steps, self._edge_ptr = outputs['classified_edges'], batch.edge_ptr

    def __len__(self):
        return len(self._edge_ptr) - 1

    def __getitem__(self, g):
        if isinstance(g, slice):
            return [self[i] for i in range(*g.indices(len(self)))]
        if g < 0:
            g += len(self)
        if not 0 <= g < len(self):
            raise IndexError(g)
        lo, hi = self._edge_ptr[g], self._edge_ptr[g + 1]
        return [t[lo:hi] for t in self._steps]

    def __iter__(self):
        return (self[g] for g in range(len(self)))

    def __eq__(self, other):
        return list(self) == other if isinstance(other, list) else NotImplemented


def split_logits(outputs, batch):
    """Per-graph views of each classified step: sequence over graphs of list over steps of Tensor[e_g, 1]."""
    return PerGraphLogits(outputs, batch)


def broadcast_weights(model, src=0, group=None):
    """Every rank receives rank `src`'s parameters and buffers: the whole state_dict travels as one flat buffer per
    dtype (268 145 fp32 values = 1.07 MB for the shipped configs, plus BatchNorm's int64 counter) -- ONE RCCL broadcast
    over xGMI on the "nccl" backend -- and is copied INTO this rank's own tensors.  After it `state_dict()` is rank
    `src`'s on every rank and the packed kernel blob is rebuilt from it by the ordinary packer (host and device packers
    are byte-identical, tests/test_gpu_parity.py), so a later `.eval()` / `.train()` / `.to()` that drops the packed
    cache cannot resurrect rank-local weights.  Issued once at start-up or after load_state_dict on `src`."""
    import torch.distributed as dist
    sd = model.state_dict()
    dev = next(model.parameters()).device
    via_host = dist.get_backend(group) != "nccl" and dev.type == "cuda"  # rehearsal backends (gloo) move host tensors
    by_dtype = {}
    for name, t in sd.items():
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype in sorted(by_dtype, key=str):
            tensors = by_dtype[dtype]
            flat = torch.cat([t.reshape(-1) for t in tensors])
            if via_host:
                flat = flat.cpu()
            dist.broadcast(flat, src=src, group=group)
            if dist.get_rank(group) != src:
                flat = flat.to(dev)
                off = 0
                for t in tensors:
                    t.copy_(flat[off:off + t.numel()].view(t.shape))
                    off += t.numel()
    model.invalidate_packed_weights()


def broadcast_packed_weights(model, src=0, group=None):
    """broadcast_weights, then pack on this rank; returns the packed blob (on the module's device, or on the host for
    a CPU module, where only the packer -- no kernel -- can run)."""
    broadcast_weights(model, src=src, group=group)
    dev = next(model.parameters()).device
    if dev.type == "cuda":
        return model._packed_weights(dev)
    return model.pack_weights_host()


def shard_batch(graphs, rank, world):
    """This rank's share of `graphs` as ONE disjoint union, built once (inputs then stay resident in HBM; the reference
    builds its union in the DataLoader, outside the model call, inference.py:279).  `graphs` is any sequence that can be
    sliced by this rank's range -- a list that is identical on every rank, or a lazy sequence that only materialises
    the graphs asked for.  Returns (lo, hi, GraphBatch or None for an empty share)."""
    lo, hi = shard_range(len(graphs), rank, world)
    if hi == lo:
        return lo, hi, None
    return lo, hi, union_graphs(graphs[lo:hi])


def forward_sharded(model, graphs, rank, world, batch=None, graphed=None):
    """Run this rank's share of `graphs` (see shard_batch; pass the `batch` it returned to reuse a union that is already
    resident instead of concatenating again).  No collective: every rank runs the ordinary single-GPU forward on its own
    union.  Returns (lo, hi, per-graph logits: list over this rank's graphs of list over classified steps).

    `graphed`: a gnn_cca_amd.inference.GraphedForward of `model` -- with a resident `batch` the rank's forward is then ONE replay of a HIP
    graph captured on that union's own tensors (`GraphedForward.block([batch], adopt_inputs=True)`: no copies; the producer overwrites
    the union in place between replays), the same kernels with the same arguments as the eager call, bit for bit its logits.  The
    returned views are the graph's static outputs: valid until the next replay."""
    if batch is None:
        lo, hi, batch = shard_batch(graphs, rank, world)
    else:
        lo, hi = shard_range(len(graphs), rank, world)
        if len(batch.node_ptr) - 1 != hi - lo:
            raise ValueError(f"batch holds {len(batch.node_ptr) - 1} graphs, this rank's share is [{lo}, {hi})")
    if batch is None:
        return lo, hi, []
    if graphed is not None:
        out = graphed.block([batch], adopt_inputs=True).replay()[0]
    else:
        out = model(batch)
    return lo, hi, split_logits(out, batch)

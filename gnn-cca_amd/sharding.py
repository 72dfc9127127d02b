"""Multi-GPU layer of the path: graph-level sharding, one process per GPU.

Frame graphs are independent (disjoint union in Batch.from_data_list, inference.py:279; no cross-graph edges) and the
weights are shared and read-only in eval mode, so the path shards by GRAPH with no data-path collective:

  * rank r of W owns graphs [lo, hi) of the batch (`shard_range`), builds its own disjoint union with local node
    numbering (`union_graphs`) and runs the ordinary single-GPU forward on it;
  * the only collective is a one-time broadcast of the packed weight blob (2.66 MB for the shipped configs) from rank 0
    over RCCL/xGMI (`broadcast_packed_weights`; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

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


def split_logits(outputs, batch):
    """Per-graph views of each classified step: list over graphs of list over steps of Tensor[e_g, 1]."""
    per_graph = []
    for g in range(len(batch.edge_ptr) - 1):
        lo, hi = batch.edge_ptr[g], batch.edge_ptr[g + 1]
        per_graph.append([t[lo:hi] for t in outputs['classified_edges']])
    return per_graph


def broadcast_packed_weights(model, src=0, group=None):
    """Rank `src` packs its state_dict into the kernel blob; every rank receives it and installs it.  One collective,
    (2.66 MB for the shipped configs: fp32 weights plus the three bf16 planes of the first encoder layer), issued once
    at start-up or after load_state_dict."""
    import torch.distributed as dist
    dev = next(model.parameters()).device
    if dist.get_backend(group) != "nccl" and dev.type == "cuda":
        # rehearsal backends (gloo) move host tensors: broadcast the host blob, then upload
        import ctypes as C
        from . import _native as nat
        if dist.get_rank(group) == src:
            host = model.pack_weights_host()
        else:
            host = torch.empty(nat.lib().gnncca_packed_weights_bytes(C.byref(model.native_dims())), dtype=torch.uint8)
        dist.broadcast(host, src=src, group=group)
        blob = host.to(dev)
        model.set_packed_weights(blob)
        return blob
    if dist.get_rank(group) == src:
        blob = model._pack_weights_device(dev) if dev.type == "cuda" else None  # packed by one kernel, no host copy
        if blob is None:
            blob = model.pack_weights_host().to(dev)
    else:
        import ctypes as C
        from . import _native as nat
        nbytes = nat.lib().gnncca_packed_weights_bytes(C.byref(model.native_dims()))
        if nbytes == 0:
            nat.check(nat.lib().gnncca_supported(C.byref(model.native_dims())), "MOTMPNet configuration")
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(blob, src=src, group=group)
    model.set_packed_weights(blob)
    return blob


def forward_sharded(model, graphs, rank, world):
    """Run this rank's share of `graphs` (a list as for union_graphs, identical on every rank or at least indexable
    by this rank's range).  Returns (lo, hi, per-graph logits)."""
    lo, hi = shard_range(len(graphs), rank, world)
    if hi == lo:
        return lo, hi, []
    batch = union_graphs(graphs[lo:hi])
    out = model(batch)
    return lo, hi, split_logits(out, batch)

"""Whole-step HIP-graph capture for training through the HIP path.

One training iteration of the reference (`train.py:454-494`: forward, loss, `loss.backward()`, `optimizer.step()`) is,
on this path, ~90 small kernel launches -- the MPN forward and backward, the on-GPU weight repack, and PyTorch's own
loss / autograd / optimizer kernels -- and is bound by the host enqueueing them (0.75 ms for a 64-frame batch whose
kernels take 0.4 ms).  Every one of those launches goes to the current stream and none synchronises, so the whole
iteration can be captured ONCE into a HIP graph and replayed: `GraphedTrainStep` does that per input shape.

    step = GraphedTrainStep(model, optimizer, lambda outputs, labels: sum(crit(t.view(-1), labels)
                                                                        for t in outputs['classified_edges']))
    for data, labels in loader:
        loss = step(data, labels)          # tensor; valid until the next call with the same shapes

Semantics: every call is exactly one optimizer step on the given batch.  The first `warmup` calls with a new
(N, E) shape run eagerly (the allocator and the workspace settle), the next one is captured while it runs, later ones
replay the graph after copying the batch into the captured input buffers.  Optimizers must be capture-safe (SGD is;
Adam needs `capturable=True`).
"""
import torch


class _Batch:
    pass


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, warmup=3, max_graphs=32):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.warmup, self.max_graphs = int(warmup), int(max_graphs)
        self._seen = {}     # shape key -> eager calls so far
        self._graphs = {}   # shape key -> (graph, static batch, static labels, static loss)

    def _eager(self, data, labels):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(data), labels)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, data, labels):
        if not data.x.is_cuda:
            raise RuntimeError("GraphedTrainStep runs on MI355X only")
        key = (tuple(data.x.shape), tuple(data.edge_index.shape), tuple(data.edge_attr.shape), tuple(labels.shape),
               data.x.dtype, data.edge_attr.dtype, labels.dtype)
        entry = self._graphs.get(key)
        if entry is not None:
            graph, sb, sl, loss = entry
            sb.x.copy_(data.x, non_blocking=True)
            sb.edge_index.copy_(data.edge_index, non_blocking=True)
            sb.edge_attr.copy_(data.edge_attr, non_blocking=True)
            sl.copy_(labels, non_blocking=True)
            graph.replay()
            return loss
        n = self._seen.get(key, 0)
        if n < self.warmup or len(self._graphs) >= self.max_graphs:
            self._seen[key] = n + 1
            return self._eager(data, labels)
        # capture this call: the step runs (once) as part of the capture's own execution below
        sb = _Batch()
        sb.x, sb.edge_index, sb.edge_attr = data.x.clone(), data.edge_index.clone(), data.edge_attr.clone()
        sl = labels.clone()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss = self._eager(sb, sl)
        graph.replay()   # capture records, it does not execute: this replay IS the step for this call
        self._graphs[key] = (graph, sb, sl, loss)
        return loss

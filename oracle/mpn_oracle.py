"""CPU ORACLE for the GNN-CCA message-passing hot path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path (``gnn-cca_amd/``) never
imports it and has no CPU fallback.

Parity status: PINNED.  The restatement below is checked (tests/test_oracle_golden.py) against golden
vectors produced by the reference's own ``MOTMPNet`` imported from /root/reference and run on CPU
(tests/golden/make_golden.py, outputs committed as tests/golden/*.npz).  The reference repository holds
no tests, fixtures or known-answer vectors of its own for this path.

It is an op-for-op restatement of the reference algorithm (gather -> concatenate -> Linear -> ReLU ->
scatter-reduce), written against numpy (any float dtype: float32 to mirror the reference, float64 as a
high-precision yardstick) and, for the timed CPU baseline, against torch CPU ops (``TorchOracle``), which
are the very kernels the reference executes on a CPU host.

Reference lines followed (relative to /root/reference):
  models/mlp.py:4-28      MLP layer stack: Linear [+BatchNorm1d if use_batchnorm and dim != 1]
                          [+ReLU if dim != 1] [+Dropout if dropout_p is not None and dim != 1]
  models/mpn.py:103-142   MLPGraphIndependent (encoder / classifier; returns (edge, node))
  models/mpn.py:59-69     EdgeModel: cat([src, dst, edge]) -> edge_mlp
  models/mpn.py:71-101    NodeModel: cat([x[row], edge]) -> node_mlp -> aggregate by `row`
  models/mpn.py:32-54     MetaLayer: edge update first, node update sees the NEW edge features
  models/mpn.py:192-202   aggregators mean / max / sum (torch_scatter 2.0.8 semantics; empty -> 0)
  models/mpn.py:183-247   reattach flags -> input widths
  models/mpn.py:250-299   forward: encode, L steps, classify from step L - num_class_steps + 1
"""
import json

import numpy as np

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, models/mlp.py:15


# --------------------------------------------------------------------------------------------------
# Layer-stack description shared by the numpy and the torch flavour
# --------------------------------------------------------------------------------------------------
def mlp_layout(input_dim, fc_dims, dropout_p, use_batchnorm):
    """Replays models/mlp.py:10-24: returns [(seq_index_of_linear, in, out, bn_seq_index|None, relu)]."""
    assert isinstance(fc_dims, (list, tuple)), "fc_dims must be either a list or a tuple"  # mlp.py:8
    layers, idx = [], 0
    for dim in fc_dims:
        lin_idx = idx
        idx += 1
        bn_idx = None
        if use_batchnorm and dim != 1:
            bn_idx = idx
            idx += 1
        relu = dim != 1
        if relu:
            idx += 1
        if dropout_p is not None and dim != 1:
            idx += 1
        layers.append((lin_idx, input_dim, dim, bn_idx, relu))
        input_dim = dim
    return layers


def graph_independent_layout(edge_in_dim=None, node_in_dim=None, edge_out_dim=None, node_out_dim=None,
                             node_fc_dims=None, edge_fc_dims=None, dropout_p=None, use_batchnorm=None):
    """models/mpn.py:111-126."""
    node = edge = None
    if node_in_dim is not None:
        node = mlp_layout(node_in_dim, list(node_fc_dims) + [node_out_dim], dropout_p, use_batchnorm)
    if edge_in_dim is not None:
        edge = mlp_layout(edge_in_dim, list(edge_fc_dims) + [edge_out_dim], dropout_p, use_batchnorm)
    return edge, node


def model_layout(model_params, arch):
    """models/mpn.py:166-180 and 183-247 -> dict of MLP layouts keyed by state_dict prefix."""
    enc = dict(model_params["encoder_feats_dict"]["edges"])
    enc.update(model_params["encoder_feats_dict"]["nodes"][arch])  # mpn.py:167-170
    cls = model_params["classifier_feats_dict"]
    agg = model_params["node_agg_fn"]
    assert agg.lower() in ("mean", "max", "sum"), "node_agg_fn can only be 'max', 'mean' or 'sum'."  # mpn.py:193
    ra_n, ra_e = model_params["reattach_initial_nodes"], model_params["reattach_initial_edges"]
    ef, nf = (2 if ra_e else 1), (2 if ra_n else 1)
    edge_in = nf * 2 * enc["node_out_dim"] + ef * enc["edge_out_dim"]  # mpn.py:213-214
    node_in = nf * enc["node_out_dim"] + enc["edge_out_dim"]  # mpn.py:215
    em, nm = model_params["edge_model_feats_dict"], model_params["node_model_feats_dict"]
    enc_edge, enc_node = graph_independent_layout(**enc)
    cls_edge, cls_node = graph_independent_layout(**cls)
    return {
        "encoder.edge_mlp": enc_edge, "encoder.node_mlp": enc_node,
        "classifier.edge_mlp": cls_edge, "classifier.node_mlp": cls_node,
        "MPNet.edge_model.edge_mlp": mlp_layout(edge_in, em["fc_dims"], em["dropout_p"], em["use_batchnorm"]),
        "MPNet.node_model.node_mlp": mlp_layout(node_in, nm["fc_dims"], nm["dropout_p"], nm["use_batchnorm"]),
        "agg": agg, "reattach_nodes": ra_n, "reattach_edges": ra_e,
        "L": model_params["num_enc_steps"], "n_cls": model_params["num_class_steps"],
    }


# --------------------------------------------------------------------------------------------------
# numpy flavour
# --------------------------------------------------------------------------------------------------
class NumpyOracle:
    """Eval-mode forward of the reference MPN in numpy.  `sd` maps reference state_dict keys -> arrays."""

    def __init__(self, model_params, arch, sd, dtype=np.float32):
        self.lay = model_layout(model_params, arch)
        self.dtype = dtype
        self.sd = {k: np.asarray(v).astype(dtype) for k, v in sd.items() if "num_batches_tracked" not in k}

    def _mlp(self, prefix, x):
        for lin, _in, _out, bn, relu in self.lay[prefix]:
            p = f"{prefix}.fc_layers.{lin}."
            x = x @ self.sd[p + "weight"].T + self.sd[p + "bias"]  # nn.Linear
            if bn is not None:  # BatchNorm1d, eval mode: running statistics
                q = f"{prefix}.fc_layers.{bn}."
                x = (x - self.sd[q + "running_mean"]) / np.sqrt(self.sd[q + "running_var"] + self.dtype(BN_EPS))
                x = x * self.sd[q + "weight"] + self.sd[q + "bias"]
            if relu:
                x = np.maximum(x, 0)
        return x

    def _aggregate(self, flow, row, n):
        """mpn.py:195-202; accumulation in edge order like torch's CPU index_add_."""
        agg = self.lay["agg"]
        if agg == "max":  # rows that receive no element are 0 (torch_scatter.scatter_max)
            has = np.zeros(n, dtype=bool)
            has[row] = True
            full = np.full((n, flow.shape[1]), -np.inf, dtype=self.dtype)
            np.maximum.at(full, row, flow)
            return np.where(has[:, None], full, 0).astype(self.dtype)
        out = np.zeros((n, flow.shape[1]), dtype=self.dtype)
        np.add.at(out, row, flow)
        if agg == "mean":
            cnt = np.maximum(np.bincount(row, minlength=n), 1).astype(self.dtype)
            out = out / cnt[:, None]
        return out

    def forward(self, x, edge_index, edge_attr, trace=None):
        lay = self.lay
        x = np.asarray(x, dtype=self.dtype)
        e_in = np.asarray(edge_attr, dtype=self.dtype)
        row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
        n = x.shape[0]
        e = self._mlp("encoder.edge_mlp", e_in) if lay["encoder.edge_mlp"] is not None else e_in  # mpn.py:270
        h = self._mlp("encoder.node_mlp", x) if lay["encoder.node_mlp"] is not None else x
        e0, h0 = e, h
        if trace is not None:
            trace["e_enc"], trace["h_enc"] = e, h
        L, first = lay["L"], lay["L"] - lay["n_cls"] + 1  # mpn.py:277
        logits = []
        for step in range(1, L + 1):
            if lay["reattach_edges"]:
                e = np.concatenate([e0, e], axis=1)  # mpn.py:283 (initial first)
            if lay["reattach_nodes"]:
                h = np.concatenate([h0, h], axis=1)  # mpn.py:285
            e = self._mlp("MPNet.edge_model.edge_mlp", np.concatenate([h[row], h[col], e], axis=1))  # mpn.py:48,68-69
            flow = self._mlp("MPNet.node_model.node_mlp", np.concatenate([h[row], e], axis=1))  # mpn.py:97-98
            h = self._aggregate(flow, row, n)  # mpn.py:99
            if trace is not None:
                trace[f"h_step_{step}"], trace[f"e_step_{step}"] = h, e
            if step >= first:
                logits.append(self._mlp("classifier.edge_mlp", e))  # mpn.py:290-293
        if L == 0:
            logits.append(self._mlp("classifier.edge_mlp", e))  # mpn.py:295-297
        return logits


# --------------------------------------------------------------------------------------------------
# torch flavour: the ops the reference itself runs on a CPU host (used as the timed CPU baseline)
# --------------------------------------------------------------------------------------------------
class TorchOracle:
    """Same restatement on torch CPU tensors: index / cat / addmm / relu / batch_norm / index_add_ --
    op for op what models/mpn.py executes (torch_scatter.scatter_add == zeros.index_add_ on CPU)."""

    def __init__(self, model_params, arch, sd):
        import torch
        self.torch = torch
        self.lay = model_layout(model_params, arch)
        self.sd = {k: torch.as_tensor(np.asarray(v)).float() for k, v in sd.items() if "num_batches_tracked" not in k}

    def _mlp(self, prefix, x):
        torch = self.torch
        for lin, _in, _out, bn, relu in self.lay[prefix]:
            p = f"{prefix}.fc_layers.{lin}."
            x = torch.nn.functional.linear(x, self.sd[p + "weight"], self.sd[p + "bias"])
            if bn is not None:
                q = f"{prefix}.fc_layers.{bn}."
                x = torch.nn.functional.batch_norm(x, self.sd[q + "running_mean"], self.sd[q + "running_var"],
                                                   self.sd[q + "weight"], self.sd[q + "bias"], False, 0.0, BN_EPS)
            if relu:
                x = torch.relu_(x)
        return x

    def _aggregate(self, flow, row, n):
        torch = self.torch
        agg = self.lay["agg"]
        if agg == "max":
            return _scatter_max_first(torch, flow, row, n)
        out = torch.zeros(n, flow.shape[1]).index_add_(0, row, flow)
        if agg == "mean":
            cnt = torch.zeros(n).index_add_(0, row, torch.ones(row.numel())).clamp_(min=1)
            out = out / cnt.view(-1, 1)
        return out

    def forward(self, x, edge_index, edge_attr):
        torch = self.torch
        lay = self.lay
        with torch.no_grad():
            x = torch.as_tensor(x).float()
            e = torch.as_tensor(edge_attr).float()
            edge_index = torch.as_tensor(edge_index).long()
            row, col = edge_index[0], edge_index[1]
            n = x.shape[0]
            if lay["encoder.edge_mlp"] is not None:
                e = self._mlp("encoder.edge_mlp", e)
            h = self._mlp("encoder.node_mlp", x) if lay["encoder.node_mlp"] is not None else x
            e0, h0 = e, h
            L, first = lay["L"], lay["L"] - lay["n_cls"] + 1
            logits = []
            for step in range(1, L + 1):
                if lay["reattach_edges"]:
                    e = torch.cat((e0, e), dim=1)
                if lay["reattach_nodes"]:
                    h = torch.cat((h0, h), dim=1)
                e = self._mlp("MPNet.edge_model.edge_mlp", torch.cat([h[row], h[col], e], dim=1))
                flow = self._mlp("MPNet.node_model.node_mlp", torch.cat([h[row], e], dim=1))
                h = self._aggregate(flow, row, n)
                if step >= first:
                    logits.append(self._mlp("classifier.edge_mlp", e))
            if L == 0:
                logits.append(self._mlp("classifier.edge_mlp", e))
            return logits


# --------------------------------------------------------------------------------------------------
# fixture helpers
# --------------------------------------------------------------------------------------------------
def load_case(path):
    """Load a tests/golden/*.npz case -> (model_params, arch, state_dict, arrays)."""
    import os
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["params_json"]))
    sd = {}
    if "weights_file" in z.files:
        w = np.load(os.path.join(os.path.dirname(path), str(z["weights_file"])), allow_pickle=False)
        sd.update({k[4:]: w[k] for k in w.files if k.startswith("sd::")})
    sd.update({k[4:]: z[k] for k in z.files if k.startswith("sd::")})
    arrays = {k: z[k] for k in z.files if not k.startswith("sd::") and k not in ("params_json", "weights_file")}
    return meta["model_params"], meta["arch"], sd, arrays


# --------------------------------------------------------------------------------------------------
# train-mode flavour with autograd (row N3): the reference's forward under torch autograd
# --------------------------------------------------------------------------------------------------
# ---- train-mode Dropout: numpy twin of the device-side mask generator (gnn-cca_amd/csrc/common.cuh: drop_hash / drop_scale) ----------
DROP_ENC_NODE1, DROP_ENC_NODE2, DROP_ENC_EDGE, DROP_EDGE_STEP, DROP_NODE_STEP, DROP_CLS = 1, 2, 3, 16, 48, 80
_M64 = (1 << 64) - 1


def drop_stream(base, li):
    """Dropout stream of layer `li` of an MLP call whose first layer has stream `base` (csrc/train_generic.cuh: drop_stream)."""
    if li == 0:
        return base
    if base == DROP_ENC_NODE1 and li == 1:
        return DROP_ENC_NODE2
    return base + 256 * li


def dropout_scale(seed, stream, rows, cols, p):
    """[rows, cols] float32 array of 0 (dropped) or 1 / (1 - p) (kept) for the activation tensor `stream` (DROP_* + step):
    element idx = row * cols + col is kept iff the hash of (seed, stream, idx) maps to u >= p.  Same integer arithmetic as the
    kernels (uint64 wrap-around), so forward, backward and this oracle agree on every mask."""
    with np.errstate(over="ignore"):
        idx = np.arange(rows * cols, dtype=np.uint64)
        key = np.uint64((int(seed) & _M64) ^ ((int(stream) * 0xD1B54A32D192ED03) & _M64))
        x = idx * np.uint64(0x9E3779B97F4A7C15) + key
        x ^= x >> np.uint64(32)
        x *= np.uint64(0xD6E8FEB86659FD93)
        x ^= x >> np.uint64(32)
        x *= np.uint64(0xD6E8FEB86659FD93)
        x ^= x >> np.uint64(32)
    u = ((x & np.uint64(0xFFFFFFFF)) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = u >= np.float32(p)
    return (keep.astype(np.float32) * (np.float32(1.0) / (np.float32(1.0) - np.float32(p)))).reshape(rows, cols)


_SCATTER_MAX_FN = None


def _scatter_max_first(torch, src, index, dim_size):
    """torch_scatter 2.0.8's scatter_max(...)[0] with ITS backward (models/mpn.py:199; package pinned at env_gnn.yml:97, absent
    here, restated from its CPU kernel): per-segment maximum, 0 for empty segments; the gradient goes, whole, to the FIRST
    source row that attains the maximum (the reducer updates on a strict `>` walking the rows in order; backward is
    grad_in.scatter_(0, arg, grad_out)).  torch's amax backward would split it evenly among tied rows instead."""
    global _SCATTER_MAX_FN
    if _SCATTER_MAX_FN is None:
        class ScatterMaxFirst(torch.autograd.Function):
            @staticmethod
            def forward(ctx, src, index, dim_size):
                idx = index.view(-1, 1).expand_as(src)
                out = torch.zeros(dim_size, src.shape[1], dtype=src.dtype).scatter_reduce(0, idx, src, reduce="amax", include_self=False)
                e = src.shape[0]
                rows = torch.arange(e).view(-1, 1).expand_as(src)
                cand = torch.where(src == out[index], rows, torch.full_like(rows, e))
                arg = torch.full(out.shape, e, dtype=torch.long).scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
                ctx.save_for_backward(arg)
                ctx.e = e
                return out

            @staticmethod
            def backward(ctx, grad_out):
                (arg,) = ctx.saved_tensors
                grad_src = torch.zeros(ctx.e + 1, grad_out.shape[1], dtype=grad_out.dtype)
                grad_src.scatter_(0, arg, grad_out)   # empty segments carry arg == e: the extra row swallows them
                return grad_src[:ctx.e], None, None

        _SCATTER_MAX_FN = ScatterMaxFirst
    return _SCATTER_MAX_FN.apply(src, index, dim_size)


class TorchTrainOracle(TorchOracle):
    """TorchOracle with gradients: parameters are autograd leaves, BatchNorm1d runs in TRAIN mode (batch statistics,
    running buffers updated with momentum 0.1 as torch.nn.BatchNorm1d does, models/mlp.py:15).  Dropout (models/mlp.py:20-21:
    after the ReLU of every layer wider than 1) is applied with the masks of `dropout_scale` when `dropout` =
    dict(p_enc=, p_edge=, p_node=, p_cls=, seed=) is given -- the mask generator the HIP kernels use.
    `loss_and_grads` forms the loss exactly as train.py:80-97 does with LOSS NAME 'BCE' (sum over the classified
    steps of BCEWithLogitsLoss(reduction='mean'))."""

    def __init__(self, model_params, arch, sd, dropout=None):
        super().__init__(model_params, arch, sd)
        self.buffers = {k: v.clone() for k, v in self.sd.items() if "running_" in k}
        self.sd = {k: (v.clone().requires_grad_(True) if "running_" not in k else v) for k, v in self.sd.items()}
        self.dropout = dropout

    def _drop(self, x, stream, p):
        if not self.dropout or p <= 0:
            return x
        return x * self.torch.from_numpy(dropout_scale(self.dropout["seed"], stream, x.shape[0], x.shape[1], p))

    def _mlp(self, prefix, x, base=None, drop_p=0.0):
        """`base`: the dropout stream of this call's first layer (layer li draws from drop_stream(base, li)); None: no dropout."""
        torch = self.torch
        for li, (lin, _in, _out, bn, relu) in enumerate(self.lay[prefix]):
            p = f"{prefix}.fc_layers.{lin}."
            x = torch.nn.functional.linear(x, self.sd[p + "weight"], self.sd[p + "bias"])
            if bn is not None:
                q = f"{prefix}.fc_layers.{bn}."
                x = torch.nn.functional.batch_norm(x, self.buffers[q + "running_mean"], self.buffers[q + "running_var"],
                                                   self.sd[q + "weight"], self.sd[q + "bias"], True, 0.1, BN_EPS)
            if relu:
                x = torch.relu(x)
                if base is not None:
                    x = self._drop(x, drop_stream(base, li), drop_p)   # nn.Dropout sits behind the ReLU (only layers wider than 1 have either)
        return x

    def forward(self, x, edge_index, edge_attr):
        torch = self.torch
        lay = self.lay
        dr = self.dropout or {}
        p_enc, p_edge, p_node, p_cls = (dr.get(k, 0.0) for k in ("p_enc", "p_edge", "p_node", "p_cls"))
        x = torch.as_tensor(x).float()
        e = torch.as_tensor(edge_attr).float()
        edge_index = torch.as_tensor(edge_index).long()
        row, col = edge_index[0], edge_index[1]
        n = x.shape[0]
        if lay["encoder.edge_mlp"] is not None:
            e = self._mlp("encoder.edge_mlp", e, DROP_ENC_EDGE, p_enc)
        h = self._mlp("encoder.node_mlp", x, DROP_ENC_NODE1, p_enc) if lay["encoder.node_mlp"] is not None else x
        e0, h0 = e, h
        L, first = lay["L"], lay["L"] - lay["n_cls"] + 1
        logits = []
        for step in range(1, L + 1):
            if lay["reattach_edges"]:
                e = torch.cat((e0, e), dim=1)
            if lay["reattach_nodes"]:
                h = torch.cat((h0, h), dim=1)
            e = self._mlp("MPNet.edge_model.edge_mlp", torch.cat([h[row], h[col], e], dim=1), DROP_EDGE_STEP + step, p_edge)
            flow = self._mlp("MPNet.node_model.node_mlp", torch.cat([h[row], e], dim=1), DROP_NODE_STEP + step, p_node)
            h = self._aggregate(flow, row, n)
            if step >= first:
                logits.append(self._mlp("classifier.edge_mlp", e, DROP_CLS + len(logits), p_cls))
        if L == 0:
            logits.append(self._mlp("classifier.edge_mlp", e, DROP_CLS, p_cls))
        return logits

    def loss_and_grads(self, x, edge_index, edge_attr, labels):
        torch = self.torch
        logits = self.forward(x, edge_index, edge_attr)
        lab = torch.as_tensor(labels).float()
        crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
        loss = sum(crit(t.view(-1), lab) for t in logits)
        keys = [k for k, v in self.sd.items() if v.requires_grad]
        grads = torch.autograd.grad(loss, [self.sd[k] for k in keys], allow_unused=True)
        return float(loss), [t.detach() for t in logits], {k: (g if g is not None else torch.zeros_like(self.sd[k])) for k, g in zip(keys, grads)}

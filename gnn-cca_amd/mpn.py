"""Python host side of the MI355X message-passing path: a drop-in for the reference's ``models/mpn.py``.

Same names, constructor arguments, ``forward(data)`` contract and ``state_dict`` keys as
``models/mpn.py:144-299`` (``MOTMPNet``) and its helper classes, so ``main.py`` / ``inference.py`` can call it
unchanged (``outputs = mpn_model(data_batch)``, inference.py:283).  What differs is where the arithmetic runs:
``forward`` packs the parameters once into an HBM blob and calls ``gnncca_mpn_forward`` (include/gnncca_mpn.h),
which enqueues the hand-written gfx950 kernels on torch's current HIP stream.  torch is used for device memory,
streams and (in sharding.py) torch.distributed only.

There is deliberately no CPU / eager-torch fallback: off-GPU tensors or a missing ``libgnncca_mpn.so`` raise.
"""
import ctypes as C

import torch
from torch import nn

from . import _native as nat
from .mlp import MLP


# ---- stand-alone calls of the sub-modules (the reference allows them; MOTMPNet.forward itself runs fused) ---------------------------
def _standalone_check(mod, what, *tensors):
    if mod.training:
        raise RuntimeError(f"{what} called on its own runs in eval mode only (train through MOTMPNet: its autograd bridge covers "
                           "the whole forward); call .eval() first")
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("gnn_cca_amd runs on MI355X only: move the module and its inputs to the GPU (there is no CPU fallback)")
    return next(t for t in tensors if t is not None).device


def _f32(t, what):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what} must be float32, got {t.dtype}")   # the reference's fp32 Linear layers raise as well
    return t.detach().contiguous()


def standalone_mlp(mlp, x):
    """MLP.forward(input) on its own (models/mlp.py:26-28): gnncca_mlp_eval."""
    dev = _standalone_check(mlp, "MLP", x)
    x = _f32(x, "input")
    if x.dim() != 2 or (mlp.plan and x.shape[1] != mlp.plan[0][0]):
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} into Linear({mlp.plan[0][0] if mlp.plan else '?'}, ...))")
    if not mlp.plan:
        return x
    lib = nat.lib()
    desc = nat.Mlp()
    _fill_mlp(desc, mlp)
    params = mlp.native_params()
    if any(p.device != dev for p in params):
        raise RuntimeError("module and input are on different devices")
    rows = x.shape[0]
    out = torch.empty((rows, mlp.plan[-1][1]), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.gnncca_mlp_eval_workspace_bytes(C.byref(desc), rows) + 256, dtype=torch.uint8, device=dev)
    pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
    with torch.cuda.device(dev):
        st = lib.gnncca_mlp_eval(C.byref(desc), pp, len(params), x.data_ptr(), rows, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _raw_stream(dev))
    nat.check(st, "gnncca_mlp_eval")
    return out


def _gather_cat(parts, rows, dev):
    """cat([t[idx] or t for (t, idx) in parts], dim=1) by gnncca_gather_cat (no torch kernel); idx: int64 [rows] or None."""
    parts = [(t, i) for t, i in parts if t is not None]
    assert 1 <= len(parts) <= 3
    for t, i in parts:
        if i is not None and i.dtype != torch.int64:
            raise RuntimeError("tensors used as indices must be long")   # the reference's x[row] raises the same way
    parts += [(None, None)] * (3 - len(parts))
    width = sum(t.shape[1] for t, _ in parts if t is not None)
    out = torch.empty((rows, width), dtype=torch.float32, device=dev)
    args = []
    for t, i in parts:
        args += [t.data_ptr() if t is not None else None, i.data_ptr() if i is not None else None,
                 t.shape[1] if t is not None else 0, t.shape[0] if t is not None else 0]
    with torch.cuda.device(dev):
        st = nat.lib().gnncca_gather_cat(*args, rows, out.data_ptr(), _raw_stream(dev))
    nat.check(st, "gnncca_gather_cat")
    return out


class _Replayed(nn.Module):
    """Base of the three top-level containers (encoder, MPNet, classifier).  They hold parameters; the arithmetic of a
    MOTMPNet forward runs fused inside libgnncca_mpn.so.  When a caller has registered forward hooks on one of them (the way
    per-step latents are usually tapped from the reference), MOTMPNet.forward runs the traced native forward and then
    REPLAYS the reference's call sequence (models/mpn.py:266-297) through ``__call__`` of these containers: each call returns
    the tensors the fused kernels produced for it, so hooks see the same inputs and outputs as on the reference -- in eval mode
    the latents of the traced forward, in train mode the latents the training forward saved for its backward (train-mode values:
    BatchNorm with batch statistics, Dropout applied; the classifier outputs are the autograd-connected logits, the latents are
    detached views).  A container called on its own, outside MOTMPNet.forward, evaluates itself with the stand-alone entry points
    (eval mode)."""

    def _take_replayed(self):
        queue = getattr(self, '_replay_queue', None)
        return queue.pop(0) if queue else None


class MetaLayer(_Replayed):
    """models/mpn.py:10-57 (``edge_model`` / ``node_model`` children)."""

    def __init__(self, edge_model=None, node_model=None):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model

    def forward(self, x, edge_index, edge_attr):
        got = self._take_replayed()
        if got is not None:
            return got                                  # (x, edge_attr), models/mpn.py:54
        dev = _standalone_check(self, "MetaLayer", x, edge_index, edge_attr)
        x, edge_attr = _f32(x, "x"), _f32(edge_attr, "edge_attr")
        row, col = edge_index[0].contiguous(), edge_index[1].contiguous()   # mpn.py:44
        e = edge_attr.shape[0]
        src = _gather_cat([(x, row)], e, dev)           # x[row], x[col] (mpn.py:48)
        dst = _gather_cat([(x, col)], e, dev)
        edge_attr = self.edge_model(src, dst, edge_attr)
        x = self.node_model(x, edge_index, edge_attr)   # mpn.py:52
        return x, edge_attr


class EdgeModel(nn.Module):
    """models/mpn.py:59-69: owns ``edge_mlp`` (input = cat[source, target, edge_attr])."""

    def __init__(self, edge_mlp):
        super().__init__()
        self.edge_mlp = edge_mlp

    def forward(self, source, target, edge_attr):
        dev = _standalone_check(self, "EdgeModel", source, target, edge_attr)
        out = _gather_cat([(_f32(source, "source"), None), (_f32(target, "target"), None), (_f32(edge_attr, "edge_attr"), None)],
                          source.shape[0], dev)         # mpn.py:68
        return self.edge_mlp(out)


class NodeModel(nn.Module):
    """models/mpn.py:71-101: owns ``node_mlp`` (input = cat[x[row], e']) and the aggregator applied over ``row``."""

    def __init__(self, node_mlp, node_agg_fn):
        super().__init__()
        self.node_mlp = node_mlp
        self.node_agg_fn = node_agg_fn

    def forward(self, x, edge_index, edge_attr):
        dev = _standalone_check(self, "NodeModel", x, edge_index, edge_attr)
        x, edge_attr = _f32(x, "x"), _f32(edge_attr, "edge_attr")
        if edge_index.dtype != torch.int64:
            raise RuntimeError("tensors used as indices must be long")
        edge_index = edge_index.contiguous()
        n, e = x.shape[0], edge_attr.shape[0]
        flow = self.node_mlp(_gather_cat([(x, edge_index[0]), (edge_attr, None)], e, dev))   # mpn.py:97-98
        lib = nat.lib()
        out = torch.empty((n, flow.shape[1]), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.gnncca_aggregate_workspace_bytes(n, e) + 256, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            st = lib.gnncca_aggregate(flow.data_ptr(), edge_index.data_ptr(), n, e, flow.shape[1], nat.AGG[self.node_agg_fn], out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _raw_stream(dev))
        nat.check(st, "gnncca_aggregate")
        return out                                       # mpn.py:99


class MLPGraphIndependent(_Replayed):
    """models/mpn.py:103-142: an optional node MLP and an optional edge MLP."""

    def __init__(self, edge_in_dim=None, node_in_dim=None, edge_out_dim=None, node_out_dim=None,
                 node_fc_dims=None, edge_fc_dims=None, dropout_p=None, use_batchnorm=None):
        super().__init__()
        self.node_mlp = None
        self.edge_mlp = None
        if node_in_dim is not None:  # node MLP is created first (parameter-init RNG order of the reference)
            self.node_mlp = MLP(node_in_dim, list(node_fc_dims) + [node_out_dim], dropout_p, use_batchnorm)
        if edge_in_dim is not None:
            self.edge_mlp = MLP(edge_in_dim, list(edge_fc_dims) + [edge_out_dim], dropout_p, use_batchnorm)

    def forward(self, edge_feats=None, nodes_feats=None):
        got = self._take_replayed()
        if got is not None:
            return got                                   # (edge_out, node_out): edge first, models/mpn.py:142
        node_out = self.node_mlp(nodes_feats) if self.node_mlp is not None and nodes_feats is not None else nodes_feats   # mpn.py:130-140
        edge_out = self.edge_mlp(edge_feats) if self.edge_mlp is not None and edge_feats is not None else edge_feats
        return edge_out, node_out


def _raw_stream(device):
    """hipStream_t of torch's current stream on `device` as an int (the fast path of torch.cuda.current_stream(...)
    .cuda_stream: no Python Stream object; it is called once per forward on a path whose whole host cost is ~20 us)."""
    return torch._C._cuda_getCurrentRawStream(device.index)


class _HotState:
    """Per-forward mutable state kept OFF nn.Module.__setattr__ (which costs ~2 us per assignment)."""
    __slots__ = ("workspace", "workspaces", "last_workspace_bytes", "n_out", "ws_shape", "ws_bytes")

    def __init__(self):
        self.workspace, self.workspaces, self.last_workspace_bytes = None, {}, 0
        self.n_out, self.ws_shape, self.ws_bytes = -1, None, 0


class _MPNTrainFunction(torch.autograd.Function):
    """Autograd bridge for train mode: forward = the traced HIP forward (it saves the latents the
    backward needs), backward = gnncca_mpn_backward.  Gradients flow to the module's parameters only (the reference
    computes the node features under torch.no_grad(), train.py:248-253, so d/dx is never asked for)."""

    @staticmethod
    def forward(ctx, module, x, edge_index, edge_attr, *params):
        trace = {}
        drop, seed = module._dropout_for_this_call(x.device)   # (nat.Dropout or None, the device seed word it points at)
        with torch.no_grad():
            out = module._forward_native(x, edge_index, edge_attr, trace, dropout=drop)
            bn_stat = torch.zeros(1, dtype=torch.float32, device=x.device)
            bn = module._classifier_batchnorm()
            if bn is not None and edge_index.shape[1] > 0:
                # BatchNorm1d in train mode: the logits are recomputed from the saved edge latents with batch statistics
                lib, d = nat.lib(), module.native_dims()
                n_out, c1, e = out.shape[0], bn.num_features, edge_index.shape[1]
                bn_stat = torch.empty((n_out, c1, 2), dtype=torch.float32, device=x.device)
                scratch = torch.empty(2 * c1, dtype=torch.float64, device=x.device)
                pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
                with torch.cuda.device(x.device):
                    st = lib.gnncca_classifier_train_dropout(C.byref(d), pp, len(params), trace['e_steps'].data_ptr(), e,
                                                             scratch.data_ptr(), bn_stat.data_ptr(), out.data_ptr(),
                                                             C.byref(drop) if drop is not None else None, _raw_stream(x.device))
                nat.check(st, "gnncca_classifier_train")
                bn.num_batches_tracked += n_out  # one BatchNorm call per classified step (models/mpn.py:292)
        if module._containers_hooked():   # forward hooks on encoder / MPNet / classifier: MOTMPNet.forward replays them from these
            module._train_latents = (trace['h_enc'], trace['e_enc'], list(trace['h_steps'].unbind(0)), list(trace['e_steps'].unbind(0)))
        ctx.module = module
        ctx.n_params = len(params)
        ctx.has_bn = bn is not None
        ctx.drop, ctx.drop_seed = drop, seed   # the backward re-derives the same masks from the same seed word
        ctx.save_for_backward(x, edge_index, edge_attr, trace['h_enc'], trace['e_enc'], trace['h_steps'], trace['e_steps'],
                              bn_stat, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, edge_index, edge_attr, h_enc, e_enc, h_steps, e_steps, bn_stat = ctx.saved_tensors[:8]
        params = ctx.saved_tensors[8:]
        module = ctx.module
        lib, d = nat.lib(), module.native_dims()
        dev = x.device
        n, e = x.shape[0], edge_index.shape[1]
        g = grad_out.reshape(grad_out.shape[0], -1).float().contiguous()
        # every gradient is a view of ONE flat buffer cleared by a single fill (GNNCCA_BWD_GRADS_ZEROED)
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        gp = (C.c_void_p * len(params))(*[t.data_ptr() for t in grads])
        ws = torch.empty(lib.gnncca_backward_workspace_bytes(C.byref(d), n, e) + 256, dtype=torch.uint8, device=dev)
        saved = nat.Trace(h_enc.data_ptr(), e_enc.data_ptr(), h_steps.data_ptr(), e_steps.data_ptr())
        with torch.cuda.device(dev):
            st = lib.gnncca_mpn_backward_train(C.byref(d), pp, len(params), x.data_ptr(), edge_index.data_ptr(),
                                               edge_attr.data_ptr(), n, e, C.byref(saved),
                                               bn_stat.data_ptr() if ctx.has_bn else None, g.data_ptr(), gp, ws.data_ptr(),
                                               ws.numel(), nat.BWD_GRADS_ZEROED,
                                               C.byref(ctx.drop) if ctx.drop is not None else None, _raw_stream(dev))
        nat.check(st, "gnncca_mpn_backward")
        return (None, None, None, None, *[gr if p.requires_grad else None for gr, p in zip(grads, params)])


class _LayerwiseTrainFunction(torch.autograd.Function):
    """Autograd bridge to the layer-by-layer training engine (gnncca_train_forward / gnncca_train_backward, csrc/train_generic.cuh):
    every legal GRAPH_NET_PARAMS in train mode -- BatchNorm with batch statistics in any MLP, Dropout, the generic family.  The tape
    (what autograd would keep) is one device buffer owned by this call."""

    @staticmethod
    def forward(ctx, module, x, edge_index, edge_attr, *params):
        lib, d = nat.lib(), module.native_dims()
        dev = x.device
        n, e = x.shape[0], edge_index.shape[1]
        n_out = lib.gnncca_num_outputs(C.byref(d))
        drop, seed = module._dropout_for_this_call(dev)
        logits = torch.zeros((n_out, e, 1), dtype=torch.float32, device=dev)
        tape = torch.empty(lib.gnncca_train_tape_bytes(C.byref(d), n, e) + 256, dtype=torch.uint8, device=dev)
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        with torch.cuda.device(dev):
            st = lib.gnncca_train_forward(C.byref(d), pp, len(params), x.data_ptr(), edge_index.data_ptr(), edge_attr.data_ptr(),
                                          n, e, tape.data_ptr(), tape.numel(), logits.data_ptr(),
                                          C.byref(drop) if drop is not None else None, _raw_stream(dev))
        nat.check(st, "gnncca_train_forward")
        if module._containers_hooked() and n > 0 and e > 0:
            L = int(module.num_enc_steps)
            offs = (C.c_int64 * (2 + 2 * L))()
            nat.check(lib.gnncca_train_tape_latents(C.byref(d), n, e, offs, 2 + 2 * L), "gnncca_train_tape_latents")

            def view(off, rows, width, passthrough):
                if off < 0:
                    return passthrough
                return tape[off:off + rows * width * 4].view(torch.float32).view(rows, width)
            module._train_latents = (view(offs[0], n, d.node_dim, x), view(offs[1], e, d.edge_dim, edge_attr),
                                     [view(offs[2 + 2 * s], n, d.node_dim, None) for s in range(L)],
                                     [view(offs[3 + 2 * s], e, d.edge_dim, None) for s in range(L)])
        module._count_batchnorm_calls(n_out if e > 0 else 0, e > 0)
        ctx.module, ctx.tape, ctx.drop, ctx.drop_seed = module, tape, drop, seed
        ctx.save_for_backward(x, edge_index, edge_attr, *params)
        return logits

    @staticmethod
    def backward(ctx, grad_out):
        x, edge_index, edge_attr = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        lib, d = nat.lib(), ctx.module.native_dims()
        dev = x.device
        n, e = x.shape[0], edge_index.shape[1]
        g = grad_out.reshape(grad_out.shape[0], -1).float().contiguous()
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        gp = (C.c_void_p * len(params))(*[t.data_ptr() for t in grads])
        with torch.cuda.device(dev):
            st = lib.gnncca_train_backward(C.byref(d), pp, len(params), x.data_ptr(), edge_index.data_ptr(), edge_attr.data_ptr(),
                                           n, e, ctx.tape.data_ptr(), ctx.tape.numel(), g.data_ptr(), gp,
                                           C.byref(ctx.drop) if ctx.drop is not None else None, _raw_stream(dev))
        nat.check(st, "gnncca_train_backward")
        return (None, None, None, None, *[gr if p.requires_grad else None for gr, p in zip(grads, params)])


def _fill_mlp(dst, mlp):
    dst.n_layers = 0 if mlp is None else len(mlp.plan)
    if dst.n_layers > nat.MAX_LAYERS:
        raise NotImplementedError(f"MLPs deeper than {nat.MAX_LAYERS} layers are not supported")
    for i in range(dst.n_layers):
        fan_in, width, has_bn, relu, _ = mlp.plan[i]
        dst.layers[i].in_dim, dst.layers[i].out_dim = fan_in, width
        dst.layers[i].has_bn, dst.layers[i].relu = int(has_bn), int(relu)


class MOTMPNet(nn.Module):
    """Drop-in for models/mpn.py:144-299.

    ``MOTMPNet(model_params, bb_encoder=None, arch=None)``; ``forward(data)`` reads ``data.x [N, node_in]``,
    ``data.edge_index [2, E] int64`` and ``data.edge_attr [E, edge_in]`` and returns
    ``{'classified_edges': [Tensor[E, 1], ...]}`` -- one tensor per classified step, last = final.
    """

    def __init__(self, model_params, bb_encoder=None, arch=None):
        super().__init__()
        self.node_cnn = bb_encoder  # stored and never used, as in the reference (mpn.py:163)
        self.model_params = model_params

        edges_params = model_params['encoder_feats_dict']['edges']
        nodes_params = model_params['encoder_feats_dict']['nodes'][arch]
        edges_params.update(nodes_params)  # in-place merge into the caller's dict, like mpn.py:167-169
        encoder_feats_dict = edges_params
        classifier_feats_dict = model_params['classifier_feats_dict']

        self.encoder = MLPGraphIndependent(**encoder_feats_dict)
        self.classifier = MLPGraphIndependent(**classifier_feats_dict)
        self.MPNet = self._build_core_MPNet(model_params=model_params, encoder_feats_dict=encoder_feats_dict)
        self.num_enc_steps = model_params['num_enc_steps']
        self.num_class_steps = model_params['num_class_steps']

        self._enc = dict(encoder_feats_dict)
        self._dims = None          # nat.MpnDims, built lazily (needs the finished module tree)
        self._packed = None        # (key, device blob)
        self._hot = _HotState()    # workspace of the last forward; (device index, stream) -> grow-only device scratch
        self._weights_dirty = True
        self._param_cache = None
        self._trainable_checked = False
        self._pack_state = None    # (device copy of the pack program, persistent blob) for the on-GPU repack
        # 'fp32' (default: bit-faithful to the reference within summation order) or 'bf16': the edge latents are kept
        # as bf16 in HBM between steps (GNNCCA_OPT_EDGE_STATE_BF16); arithmetic stays fp32
        self.edge_state_dtype = 'fp32'
        # split-bf16 products of the first encoder layer on batches of >= 4096 nodes: 6 (default, fp32-level accuracy) or 3
        # (GNNCCA_OPT_ENC_SPLIT3: ~2^-17 relative on that layer, logits measured 1.5e-7 off; the GEMM runs 19-28 % faster)
        self.encoder_products = 6
        # True: forwards over >= 4096 nodes never split K in the first encoder layer (GNNCCA_OPT_ENC_UNSPLIT), so a graph's logits are
        # bit for bit independent of the batch / shard it is computed in (default False: within rounding, <= 2e-6; the un-split kernel
        # of mid-size batches is 4-20 % slower on the encoder)
        self.encoder_unsplit = False
        # True: GNNCCA_OPT_COLUMN_RANGES -- steps 2 ... L compute the target ids from per-node column ranges derived by step 1 when every
        # node's ids form <= 2 contiguous runs (dense and cross-camera graphs).  Same bits either way; default False: measured, it
        # gains nothing on MI355X (include/gnncca_mpn.h)
        self.column_ranges = False
        # train mode: 'auto' = the fused kernels where they apply (the shipped shapes), else the layer-by-layer engine;
        # 'layerwise' / 'fused' force one (set it before the first training forward, or call .train() again)
        self.train_engine = 'auto'

    def _options(self):
        if self.edge_state_dtype not in ('fp32', 'bf16'):
            raise ValueError("edge_state_dtype must be 'fp32' or 'bf16'")
        if self.encoder_products not in (3, 6):
            raise ValueError("encoder_products must be 6 or 3")
        return (nat.OPT_EDGE_STATE_BF16 if self.edge_state_dtype == 'bf16' else 0) | \
               (nat.OPT_ENC_SPLIT3 if self.encoder_products == 3 else 0) | \
               (nat.OPT_ENC_UNSPLIT if self.encoder_unsplit else 0) | \
               (nat.OPT_COLUMN_RANGES if getattr(self, 'column_ranges', False) else 0)

    # -- construction --------------------------------------------------------------------------------------
    def _build_core_MPNet(self, model_params, encoder_feats_dict):
        node_agg_fn = model_params['node_agg_fn']
        assert node_agg_fn.lower() in ('mean', 'max', 'sum'), "node_agg_fn can only be 'max', 'mean' or 'sum'."
        self.reattach_initial_nodes = model_params['reattach_initial_nodes']
        self.reattach_initial_edges = model_params['reattach_initial_edges']
        nf = 2 if self.reattach_initial_nodes else 1
        ef = 2 if self.reattach_initial_edges else 1
        h, f = encoder_feats_dict['node_out_dim'], encoder_feats_dict['edge_out_dim']
        edge_cfg, node_cfg = model_params['edge_model_feats_dict'], model_params['node_model_feats_dict']
        # widths of the concatenations at mpn.py:68 and mpn.py:97
        edge_mlp = MLP(nf * 2 * h + ef * f, edge_cfg['fc_dims'], edge_cfg['dropout_p'], edge_cfg['use_batchnorm'])
        node_mlp = MLP(nf * h + f, node_cfg['fc_dims'], node_cfg['dropout_p'], node_cfg['use_batchnorm'])
        # The reference then builds an unused `node_mlp_old` = Linear(2h, h) + ReLU (mpn.py:241-242).  It registers nothing, but its
        # initialisation draws from torch's global generator after every real parameter exists: a seeded script (main_training.py seeds,
        # builds the model, then shuffles / samples) sees a different stream afterwards unless the same amount is drawn here.  The
        # throwaway layer is built and dropped (tests/test_boundary.py: same seed => same parameters AND same generator state as the
        # reference, pinned by tests/golden/rng_after_init.npz).
        nn.Linear(2 * h, h)
        return MetaLayer(edge_model=EdgeModel(edge_mlp), node_model=NodeModel(node_mlp, node_agg_fn.lower()))

    # -- native description ----------------------------------------------------------------------------------
    def native_dims(self):
        if self._dims is None:
            d = nat.MpnDims()
            d.abi_version = nat.ABI_VERSION
            d.node_in = self._enc['node_in_dim']
            d.edge_in = self._enc['edge_in_dim']
            d.node_dim = self._enc['node_out_dim']
            d.edge_dim = self._enc['edge_out_dim']
            d.agg = nat.AGG[self.MPNet.node_model.node_agg_fn]
            d.num_enc_steps, d.num_class_steps = int(self.num_enc_steps), int(self.num_class_steps)
            d.reattach_nodes, d.reattach_edges = int(self.reattach_initial_nodes), int(self.reattach_initial_edges)
            _fill_mlp(d.enc_node, self.encoder.node_mlp)
            _fill_mlp(d.enc_edge, self.encoder.edge_mlp)
            _fill_mlp(d.edge_mlp, self.MPNet.edge_model.edge_mlp)
            _fill_mlp(d.node_mlp, self.MPNet.node_model.node_mlp)
            _fill_mlp(d.cls_edge, self.classifier.edge_mlp)
            self._dims = d
        return self._dims

    def native_param_tensors(self):
        """Parameters and BatchNorm buffers in gnncca_pack_weights order (cached: walking the module tree costs more
        than the whole GPU forward of a small graph; `_apply` drops the cache because it replaces buffer objects)."""
        if self._param_cache is None:
            self._param_cache = self._collect_param_tensors()
        return self._param_cache

    def _collect_param_tensors(self):
        out = []
        for mlp in (self.encoder.node_mlp, self.encoder.edge_mlp, self.MPNet.edge_model.edge_mlp,
                    self.MPNet.node_model.node_mlp, self.classifier.edge_mlp):
            if mlp is not None:
                out += mlp.native_params()
        return out

    # -- weight cache ----------------------------------------------------------------------------------------
    def invalidate_packed_weights(self):
        self._weights_dirty = True

    def _apply(self, fn, *a, **k):  # .cuda() / .to() / .float()
        self._weights_dirty = True
        self._param_cache = None
        self._pack_state = None
        self._hot = _HotState()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._weights_dirty = True
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        self._weights_dirty = True  # parameters may have been updated in place while training
        self._trainable_checked = False
        return super().train(mode)

    def pack_weights_host(self):
        """state_dict -> the packed blob of gnncca_pack_weights, as a CPU uint8 tensor (what rank 0 broadcasts)."""
        lib, d = nat.lib(), self.native_dims()
        nat.check(lib.gnncca_supported(C.byref(d)), "MOTMPNet configuration")
        host = [t.detach().to("cpu", torch.float32).contiguous() for t in self.native_param_tensors()]
        ptrs = (C.c_void_p * len(host))(*[t.data_ptr() for t in host])
        nbytes = lib.gnncca_packed_weights_bytes(C.byref(d))
        blob = torch.empty(nbytes, dtype=torch.uint8)  # gnncca_pack_weights clears and fills every byte
        nat.check(lib.gnncca_pack_weights(C.byref(d), ptrs, len(host), blob.data_ptr(), nbytes), "gnncca_pack_weights")
        return blob

    def set_packed_weights(self, blob_dev):
        """Install an already packed (e.g. RCCL-broadcast) blob living on this module's device."""
        self._packed = (self._version_key(), blob_dev)
        self._weights_dirty = False

    def load_packed_blob(self, blob):
        """Install a blob that was packed EARLIER (e.g. written by `python -m gnn_cca_amd.checkpoint convert`): a host
        uint8 tensor / bytes / file path.  The header is validated against this build's layout generation and this
        module's configuration before anything reaches the GPU -- a blob is only valid for the library build and
        GRAPH_NET_PARAMS that produced it."""
        import struct
        if isinstance(blob, str):
            with open(blob, "rb") as f:
                blob = f.read()
        if isinstance(blob, (bytes, bytearray)):
            blob = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        blob = blob.detach().to("cpu", torch.uint8).contiguous()
        lib, d = nat.lib(), self.native_dims()
        expected = lib.gnncca_packed_weights_bytes(C.byref(d))
        if blob.numel() < 16:
            raise RuntimeError("packed blob: too short to hold a header")
        magic, abi, _family, total_floats = struct.unpack("<4I", blob[:16].numpy().tobytes())
        ref_magic, = struct.unpack("<I", self.pack_weights_host()[:4].numpy().tobytes())
        if magic != ref_magic or abi != nat.ABI_VERSION:
            raise RuntimeError("packed blob was written by a different build of libgnncca_mpn (layout generation "
                               f"{magic:#x}, expected {ref_magic:#x}): re-run the converter")
        if blob.numel() != expected or total_floats * 4 != expected:
            raise RuntimeError(f"packed blob is {blob.numel()} bytes; this configuration needs {expected}")
        dev = next(self.parameters()).device
        self.set_packed_weights(blob.to(dev))
        return self

    def _version_key(self):
        return sum(t._version for t in self.native_param_tensors())

    def _pack_weights_device(self, device):
        """Repack on the GPU (gnncca_pack_weights_device): no host round trip, enqueued on the current stream.
        Returns None when this configuration / parameter placement needs the host packer."""
        lib, d = nat.lib(), self.native_dims()
        params = self.native_param_tensors()
        if any(t.device != device or t.dtype != torch.float32 or not t.is_contiguous() for t in params):
            return None
        st = self._pack_state
        if st is None or st[0].device != device:
            nbytes = lib.gnncca_pack_program_bytes()
            prog = torch.zeros(nbytes, dtype=torch.uint8)
            if lib.gnncca_pack_program(C.byref(d), prog.data_ptr(), nbytes) != nat.OK:
                return None  # generic family
            blob = torch.zeros(lib.gnncca_packed_weights_bytes(C.byref(d)), dtype=torch.uint8, device=device)
            st = self._pack_state = (prog.to(device), blob)
        prog, blob = st
        ptrs = (C.c_void_p * len(params))(*[t.data_ptr() for t in params])
        with torch.cuda.device(device):
            status = lib.gnncca_pack_weights_device(C.byref(d), ptrs, len(params), prog.data_ptr(), blob.data_ptr(), blob.numel(),
                                                    _raw_stream(device))
        nat.check(status, "gnncca_pack_weights_device")
        return blob

    def _packed_weights(self, device):
        if self._weights_dirty or self._packed is None or self._packed[1].device != device \
                or self._packed[0] != self._version_key() \
                or (self.training and torch.cuda.is_current_stream_capturing()):
            # (a training step being captured into a HIP graph must contain the repack: replays run no Python)
            blob = self._pack_weights_device(device)
            self.set_packed_weights(blob if blob is not None else self.pack_weights_host().to(device))
        return self._packed[1]

    def _scratch(self, nbytes, device):
        """Grow-only HBM workspace (CSR plan, edge state, node tables) of the CURRENT STREAM: forwards of one module on
        different streams run concurrently on separate workspaces (the packed weights are shared, read-only)."""
        hot = self._hot
        key = (device.index, _raw_stream(device))
        ws = hot.workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = hot.workspaces[key] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        hot.workspace = ws  # the last one used: graph_flags() reads it
        hot.last_workspace_bytes = nbytes
        return ws

    @property
    def last_workspace_bytes(self):
        return self._hot.last_workspace_bytes

    @property
    def _workspaces(self):
        return self._hot.workspaces

    # -- forward -----------------------------------------------------------------------------------------------
    @staticmethod
    def _check_inputs(x, edge_index, edge_attr):
        """dtype contract of the reference: its fp32 Linear layers raise a RuntimeError on any other floating type
        (models/mlp.py:13 via models/mpn.py:131,137) and tensor indexing wants int64 (models/mpn.py:48).  A wrong dtype is a
        caller bug and is raised here as well instead of being converted silently; a non-contiguous VIEW of the right dtype
        is only a layout and is copied."""
        if x.dtype != torch.float32 or edge_attr.dtype != torch.float32:
            raise RuntimeError(f"expected float32 node / edge features, got x {x.dtype}, edge_attr {edge_attr.dtype} "
                               "(the module's Linear layers are float32, as in the reference)")
        if edge_index.dtype != torch.int64:
            raise RuntimeError(f"edge_index must be int64 (torch.long), got {edge_index.dtype}")
        return x.contiguous(), edge_index.contiguous(), edge_attr.contiguous()

    def _prepare(self, x, edge_index, edge_attr):
        if not (x.is_cuda and edge_index.is_cuda and edge_attr.is_cuda):
            raise RuntimeError("gnn_cca_amd.MOTMPNet runs on MI355X only: move the module and `data` to the GPU "
                               "(there is no CPU fallback)")
        lib, d = nat.lib(), self.native_dims()
        dev = x.device
        x, edge_index, edge_attr = self._check_inputs(x, edge_index, edge_attr)
        n, e = x.shape[0], edge_index.shape[1]
        if x.dim() != 2 or x.shape[1] != d.node_in or edge_index.dim() != 2 or edge_index.shape[0] != 2 \
                or edge_attr.dim() != 2 or edge_attr.shape[0] != e or edge_attr.shape[1] != d.edge_in:
            raise RuntimeError(f"shape mismatch: x {tuple(x.shape)}, edge_index {tuple(edge_index.shape)}, "
                               f"edge_attr {tuple(edge_attr.shape)} for node_in={d.node_in}, edge_in={d.edge_in}")
        blob = self._packed_weights(dev)
        hot = self._hot
        if hot.n_out < 0:
            hot.n_out = lib.gnncca_num_outputs(C.byref(d))
        logits = torch.empty((hot.n_out, e, 1), dtype=torch.float32, device=dev)
        ws = None
        if n > 0 and e > 0:
            if hot.ws_shape != (n, e):  # pure function of (dims, N, E): skip the library call while the shape repeats
                hot.ws_shape, hot.ws_bytes = (n, e), lib.gnncca_workspace_bytes(C.byref(d), n, e)
            ws_bytes = hot.ws_bytes
            if ws_bytes == 0:
                nat.check(lib.gnncca_supported(C.byref(d)), "MOTMPNet configuration")
            ws = self._scratch(ws_bytes, dev)
        return lib, d, dev, x, edge_index, edge_attr, n, e, blob, logits, ws

    def forward(self, data, trace=None):
        """See class docstring.  ``trace`` (optional dict) receives the intermediate latents for debugging.
        In train mode the outputs carry an autograd graph to the parameters (row N3)."""
        if self.training:
            return self._forward_train(data)
        if trace is None and self._containers_hooked():
            return self._forward_replayed(data)
        logits = self._forward_native(data.x, data.edge_index, data.edge_attr, trace)
        return {'classified_edges': list(logits.unbind(0))}

    def _containers_hooked(self):
        for m in (self.encoder, self.MPNet, self.classifier):
            if m._forward_hooks or m._forward_pre_hooks:
                return True
        return False

    def _forward_replayed(self, data):
        """Forward with forward hooks registered on encoder / MPNet / classifier: the traced native forward, then the
        reference's own call sequence (models/mpn.py:266-297) through the containers, each call handing back what the fused
        kernels computed for it.  Hooks on deeper modules (edge_model, node_model, the MLPs) do not fire: those calls do not
        exist as separate steps on this path."""
        trace = {}
        logits = self._forward_native(data.x, data.edge_index, data.edge_attr, trace)
        return self._replay_containers(data, list(logits.unbind(0)), trace['h_enc'], trace['e_enc'], list(trace['h_steps'].unbind(0)),
                                       list(trace['e_steps'].unbind(0)))

    def _replay_containers(self, data, logits, h_enc, e_enc, h_steps, e_steps):
        """The call sequence of models/mpn.py:266-297 through the three containers; every call returns the given tensors."""
        x, edge_index, edge_attr = data.x, data.edge_index, data.edge_attr
        L, n_cls = int(self.num_enc_steps), int(self.num_class_steps)
        enc_q, mp_q, cls_q = [(e_enc, h_enc)], [], []
        first_class_step = L - n_cls + 1
        for step in range(1, L + 1):
            mp_q.append((h_steps[step - 1], e_steps[step - 1]))
        for t in logits:
            cls_q.append((t, None))
        self.encoder._replay_queue, self.MPNet._replay_queue, self.classifier._replay_queue = enc_q, mp_q, cls_q
        try:
            latent_edge, latent_node = self.encoder(edge_attr, x)
            initial_edge, initial_node = latent_edge, latent_node
            out = []
            for step in range(1, L + 1):
                if self.reattach_initial_nodes:
                    latent_node = torch.cat((initial_node, latent_node), dim=1)
                if self.reattach_initial_edges:
                    latent_edge = torch.cat((initial_edge, latent_edge), dim=1)
                latent_node, latent_edge = self.MPNet(latent_node, edge_index, latent_edge)
                if step >= first_class_step:
                    dec_edge, _ = self.classifier(latent_edge)
                    out.append(dec_edge)
            if L == 0:
                dec_edge, _ = self.classifier(latent_edge)
                out.append(dec_edge)
        finally:
            self.encoder._replay_queue = self.MPNet._replay_queue = self.classifier._replay_queue = None
        return {'classified_edges': out}

    def _classifier_batchnorm(self):
        mlp = self.classifier.edge_mlp
        for bi in (mlp.bn_index if mlp is not None else []):
            if bi is not None:
                return mlp.fc_layers[bi]
        return None

    def _check_trainable(self):
        if self._trainable_checked:  # reset by .train() / _apply(); Dropout.p edited by hand afterwards is not re-read
            return
        lib, d = nat.lib(), self.native_dims()
        # the shipped shapes (BatchNorm nowhere or only inside the classifier, two-layer node encoder) train on the fused kernels;
        # everything else -- BatchNorm in the encoder / MPN MLPs, the generic family -- on the layer-by-layer engine
        fused_ok = lib.gnncca_backward_supported(C.byref(d)) == nat.OK
        want = getattr(self, 'train_engine', 'auto')
        if want not in ('auto', 'fused', 'layerwise'):
            raise ValueError("train_engine must be 'auto', 'fused' or 'layerwise'")
        if want == 'fused' and not fused_ok:
            raise NotImplementedError("the fused training kernels cover the shipped config shapes only; this configuration trains "
                                      "on train_engine = 'layerwise' (SURVEY.md 8f row N3)")
        self._train_path = 'fused' if (fused_ok and want != 'layerwise') else 'layerwise'
        self._dropout_ps()   # raises if the Dropout modules of one group disagree
        self._trainable_checked = True

    def _check_batchnorm_rows(self, n, e):
        """torch.nn.BatchNorm1d in train mode refuses a batch of exactly ONE row (its _verify_batch_size rejects one value per
        channel): so does this module, with torch's words.  Zero rows are not refused: a frame without edges skips the edge MLPs'
        BatchNorm calls (both engines do) and returns empty logits."""
        for mlp, rows in ((self.encoder.node_mlp, n), (self.encoder.edge_mlp, e), (self.MPNet.edge_model.edge_mlp, e),
                          (self.MPNet.node_model.node_mlp, e), (self.classifier.edge_mlp, e)):
            if mlp is None or rows != 1:
                continue
            if mlp is not self.encoder.node_mlp and mlp is not self.encoder.edge_mlp and mlp is not self.classifier.edge_mlp \
                    and int(self.num_enc_steps) == 0:
                continue   # the MPN MLPs are never called when L == 0
            for mod in mlp.fc_layers:
                if isinstance(mod, nn.BatchNorm1d):
                    raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                                     f"torch.Size([{rows}, {mod.num_features}])")

    def _count_batchnorm_calls(self, n_cls_calls, edges):
        """num_batches_tracked of every BatchNorm1d after one train-mode forward (one increment per call of its MLP)."""
        L = int(self.num_enc_steps)
        calls = ((self.encoder.node_mlp, 1), (self.encoder.edge_mlp, 1 if edges else 0),
                 (self.MPNet.edge_model.edge_mlp, L if edges else 0), (self.MPNet.node_model.node_mlp, L if edges else 0),
                 (self.classifier.edge_mlp, n_cls_calls))
        for mlp, k in calls:
            if mlp is None or k == 0:
                continue
            for mod in mlp.fc_layers:
                if isinstance(mod, nn.BatchNorm1d):
                    mod.num_batches_tracked += k

    # -- train-mode Dropout (models/mlp.py:20-21) -------------------------------------------------------------------------
    def _dropout_ps(self):
        """(p_enc, p_edge, p_node, p_cls) read from the nn.Dropout modules of the MLPs (so a p edited after construction
        counts, as on the reference).  The two encoder MLPs share one config entry and one kernel parameter."""
        def group_p(*mlps):
            ps = {float(m.p) for mlp in mlps if mlp is not None for m in mlp.fc_layers if isinstance(m, nn.Dropout)}
            if len(ps) > 1:
                raise NotImplementedError(f"different Dropout probabilities inside one MLP group ({sorted(ps)}) are not supported")
            return ps.pop() if ps else 0.0
        ps = (group_p(self.encoder.node_mlp, self.encoder.edge_mlp), group_p(self.MPNet.edge_model.edge_mlp),
              group_p(self.MPNet.node_model.node_mlp), group_p(self.classifier.edge_mlp))
        if any(not 0.0 <= q < 1.0 for q in ps):
            raise ValueError("Dropout probabilities must lie in [0, 1)")
        return ps

    def set_dropout_seed(self, seed):
        """Seed of the train-mode Dropout masks (a device word; it advances by one per training forward).  Without this call
        the first training forward draws it from torch's RNG."""
        self._drop_seed_value = int(seed) & 0x7FFFFFFFFFFFFFFF
        self._drop_seed = None

    def _dropout_for_this_call(self, device):
        ps = self._dropout_ps()
        if not any(q > 0 for q in ps):
            return None, None
        if getattr(self, '_drop_seed', None) is None or self._drop_seed.device != device:
            value = getattr(self, '_drop_seed_value', None)
            if value is None:
                value = int(torch.randint(0, 2 ** 62, (1,)).item())
            self._drop_seed = torch.tensor([value], dtype=torch.int64, device=device)
        seed = self._drop_seed.clone()   # this iteration's word: forward and backward both read it
        self._drop_seed.add_(1)          # capturable: a replayed training step draws fresh masks
        drop = nat.Dropout(ps[0], ps[1], ps[2], ps[3], seed.data_ptr())
        return drop, seed

    def _forward_train(self, data):
        self._check_trainable()
        params = self.native_param_tensors()
        if not (data.x.is_cuda and data.edge_index.is_cuda and data.edge_attr.is_cuda):
            raise RuntimeError("gnn_cca_amd.MOTMPNet runs on MI355X only: move the module and `data` to the GPU "
                               "(there is no CPU fallback)")
        x, edge_index, edge_attr = self._check_inputs(data.x, data.edge_index, data.edge_attr)
        self._check_batchnorm_rows(x.shape[0], edge_index.shape[1])
        fn = _MPNTrainFunction if self._train_path == 'fused' else _LayerwiseTrainFunction
        self._train_latents = None
        logits = fn.apply(self, x.detach(), edge_index, edge_attr.detach(), *params)
        latents, self._train_latents = self._train_latents, None
        if latents is not None:   # forward hooks on the containers: the reference's call sequence, fed from what the forward saved
            return self._replay_containers(data, list(logits.unbind(0)), *latents)
        return {'classified_edges': list(logits.unbind(0))}

    def _forward_native(self, x, edge_index, edge_attr, trace=None, dropout=None):
        """Eval-semantics forward through gnncca_mpn_forward (train-mode Dropout when `dropout` is given); returns logits
        [n_out, E, 1]."""
        lib, d, dev, x, edge_index, edge_attr, n, e, blob, logits, ws = self._prepare(x, edge_index, edge_attr)
        if ws is None:
            if trace is not None:
                L = int(self.num_enc_steps)
                trace['h_enc'] = torch.zeros((n, d.node_dim), dtype=torch.float32, device=dev)
                trace['e_enc'] = torch.zeros((e, d.edge_dim), dtype=torch.float32, device=dev)
                trace['h_steps'] = torch.zeros((L, n, d.node_dim), dtype=torch.float32, device=dev)
                trace['e_steps'] = torch.zeros((L, e, d.edge_dim), dtype=torch.float32, device=dev)
            return logits
        tr = None
        if trace is not None:
            L = int(self.num_enc_steps)
            trace['h_enc'] = torch.empty((n, d.node_dim), dtype=torch.float32, device=dev)
            trace['e_enc'] = torch.empty((e, d.edge_dim), dtype=torch.float32, device=dev)
            trace['h_steps'] = torch.empty((L, n, d.node_dim), dtype=torch.float32, device=dev)
            trace['e_steps'] = torch.empty((L, e, d.edge_dim), dtype=torch.float32, device=dev)
            tr = C.byref(nat.Trace(trace['h_enc'].data_ptr(), trace['e_enc'].data_ptr(),
                                   trace['h_steps'].data_ptr(), trace['e_steps'].data_ptr()))
        if dropout is not None:
            with torch.cuda.device(dev):
                st = lib.gnncca_mpn_forward_train(C.byref(d), blob.data_ptr(), x.data_ptr(), edge_index.data_ptr(),
                                                  edge_attr.data_ptr(), n, e, ws.data_ptr(), ws.numel(), logits.data_ptr(), tr,
                                                  C.byref(dropout), _raw_stream(dev))
            nat.check(st, "gnncca_mpn_forward_train")
            return logits
        args = (C.byref(d), blob.data_ptr(), x.data_ptr(), edge_index.data_ptr(), edge_attr.data_ptr(), n, e, ws.data_ptr(),
                ws.numel(), logits.data_ptr(), tr, self._options(), _raw_stream(dev))
        if torch._C._cuda_getDevice() == dev.index:   # the usual case: no device guard to enter and leave
            st = lib.gnncca_mpn_forward_ex(*args)
        else:
            with torch.cuda.device(dev):
                st = lib.gnncca_mpn_forward_ex(*args)
        nat.check(st, "gnncca_mpn_forward")
        return logits

    def forward_profiled(self, data):
        """Diagnostic (bench.py): same forward with a hipEvent after every kernel launch; synchronises.
        Returns (outputs, [(kernel_kind, milliseconds), ...])."""
        lib, d, dev, x, edge_index, edge_attr, n, e, blob, logits, ws = self._prepare(data.x, data.edge_index, data.edge_attr)
        if ws is None:
            return {'classified_edges': list(logits.unbind(0))}, []
        prof = nat.Profile()
        prof.options = self._options()
        with torch.cuda.device(dev):
            stream = _raw_stream(dev)
            st = lib.gnncca_mpn_forward_profiled(C.byref(d), blob.data_ptr(), x.data_ptr(), edge_index.data_ptr(),
                                                 edge_attr.data_ptr(), n, e, ws.data_ptr(), ws.numel(),
                                                 logits.data_ptr(), stream, C.byref(prof))
        nat.check(st, "gnncca_mpn_forward_profiled")
        times = [(nat.KERNEL_KINDS[prof.kind[i]], float(prof.ms[i])) for i in range(prof.count)]
        return {'classified_edges': list(logits.unbind(0))}, times

    def column_ranges_state(self):
        """Synchronises; 0 = the last forward found every node's target ids to be <= 2 contiguous runs (or never asked: fewer than two
        steps, general kernels, column_ranges left False), 1 = some node's were not and every step streamed the ids."""
        ws = self._hot.workspace
        if ws is None:
            return 0
        out = (C.c_uint32 * 2)(0, 0)
        nat.check(nat.lib().gnncca_read_graph_flags2(ws.data_ptr(), out, _raw_stream(ws.device)), "read_graph_flags2")
        return int(out[1])

    def graph_flags(self):
        """Synchronises and returns the flag word of the last forward (bit 0: unsorted rows, bit 1: bad index)."""
        ws = self._hot.workspace
        if ws is None:
            return 0
        out = C.c_uint32(0)
        dev = ws.device
        nat.check(nat.lib().gnncca_read_graph_flags(ws.data_ptr(), C.byref(out), _raw_stream(dev)), "read_graph_flags")
        return int(out.value)

#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py

What it does
------------
Imports the reference's ``models/mpn.py`` (``MOTMPNet``) unmodified from /root/reference and runs it
on CPU (fp32, eval mode, no grad) over a fixed list of cases.  For each case it stores the inputs,
the exact weights used (``state_dict``) and every observable of the forward pass: the list
``outputs['classified_edges']`` plus, for debugging, the encoder outputs and the (node, edge)
latents after each message-passing step (captured with forward hooks on ``model.encoder`` /
``model.MPNet`` -- hooks observe, they do not change the computation).

Only numbers leave this script: ``.npz`` files with inputs / weights / outputs.  No text, bytecode
or excerpt of the reference is written anywhere.

Third-party dependency not under /root/reference
------------------------------------------------
``models/mpn.py:4`` imports ``torch_scatter`` (pinned ``torch_scatter==2.0.8`` in ``env_gnn.yml:97``)
which is not installed here and cannot be fetched (no network).  The three functions the path calls
(``mpn.py:196,199,202``) are restated below from the package's published semantics and registered
as an in-process stand-in module before the import:

* ``scatter_add(src, index, dim=0, dim_size=N)``  -> zeros(N, C).index_add_(0, index, src)
* ``scatter_mean``                                -> scatter_add / clamp(count, min=1)
* ``scatter_max`` -> (values, argmax); rows that receive no element are 0 in ``values``.

Weights
-------
PyTorch default init under ``torch.manual_seed(seed)``; BatchNorm buffers/affine get non-trivial
values; then the MPN node-MLP weight and bias are scaled by ``node_mlp_scale`` (about 1/out-degree)
so that the 'sum' aggregation does not blow activations up by xN per step and an
absolute tolerance on the logits is meaningful.  The *scaled* weights are what is stored.
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ----------------------------------------------------------------------------------------------
# torch_scatter stand-in (published semantics of torch_scatter 2.0.8, see module docstring)
# ----------------------------------------------------------------------------------------------
def _install_torch_scatter_standin():
    ts = types.ModuleType("torch_scatter")

    def scatter_add(src, index, dim=0, dim_size=None):
        assert dim == 0
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
        return out.index_add_(0, index, src)

    def scatter_mean(src, index, dim=0, dim_size=None):
        out = scatter_add(src, index, dim, dim_size)
        cnt = torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp_(min=1)
        return out / cnt.view(-1, *([1] * (src.dim() - 1)))

    class _ScatterMaxFirst(torch.autograd.Function):
        """torch_scatter 2.0.8 (env_gnn.yml:97) semantics of scatter_max restated from its CPU kernel: the output is the
        per-segment maximum (0 for segments that receive nothing), `arg` the FIRST source row that attains it (the reducer
        updates on a strict `>`, walking the rows in order), and the backward hands the whole gradient to that one row
        (grad_in.scatter_(dim, arg, grad_out)) -- torch's own amax backward would split it evenly among tied rows."""

        @staticmethod
        def forward(ctx, src, index, dim_size):
            idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
            out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
            out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
            e = src.shape[0]
            rows = torch.arange(e).view(-1, *([1] * (src.dim() - 1))).expand_as(src)
            cand = torch.where(src == out[index], rows, torch.full_like(rows, e))
            arg = torch.full(out.shape, e, dtype=torch.long).scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
            ctx.save_for_backward(arg)
            ctx.e = e
            return out, arg

        @staticmethod
        def backward(ctx, grad_out, _grad_arg):
            (arg,) = ctx.saved_tensors
            grad_src = torch.zeros((ctx.e + 1,) + tuple(grad_out.shape[1:]), dtype=grad_out.dtype)
            grad_src.scatter_(0, arg, grad_out)          # empty segments carry arg == e: the extra row swallows them
            return grad_src[:ctx.e], None, None

    def scatter_max(src, index, dim=0, dim_size=None):
        assert dim == 0
        return _ScatterMaxFirst.apply(src, index, dim_size)

    ts.scatter_add, ts.scatter_mean, ts.scatter_max = scatter_add, scatter_mean, scatter_max
    sys.modules["torch_scatter"] = ts


# ----------------------------------------------------------------------------------------------
# model_params builders (the GRAPH_NET_PARAMS contract, config_inference.yaml:76-163)
# ----------------------------------------------------------------------------------------------
def make_params(node_in=2048, node_fc=(128,), node_out=32, edge_in=4, edge_fc=(), edge_out=6,
                edge_mlp_fc=(6,), node_mlp_fc=(32,), cls_fc=(4,), cls_bn=True, agg="sum", L=4, n_cls=3,
                reattach_nodes=False, reattach_edges=False, enc_bn=False, enc_dropout=0,
                mpn_bn=False, mpn_dropout=0, arch="resnet50"):
    return {
        "node_agg_fn": agg,
        "num_enc_steps": L,
        "num_class_steps": n_cls,
        "reattach_initial_nodes": reattach_nodes,
        "reattach_initial_edges": reattach_edges,
        "encoder_feats_dict": {
            "edges": {"edge_in_dim": edge_in, "edge_fc_dims": list(edge_fc), "edge_out_dim": edge_out},
            "nodes": {arch: {"node_in_dim": node_in, "node_fc_dims": list(node_fc), "node_out_dim": node_out,
                             "dropout_p": enc_dropout, "use_batchnorm": enc_bn}},
        },
        "edge_model_feats_dict": {"fc_dims": list(edge_mlp_fc), "dropout_p": mpn_dropout, "use_batchnorm": mpn_bn},
        "node_model_feats_dict": {"fc_dims": list(node_mlp_fc), "dropout_p": mpn_dropout, "use_batchnorm": mpn_bn},
        "classifier_feats_dict": {"edge_in_dim": edge_mlp_fc[-1], "edge_fc_dims": list(cls_fc), "edge_out_dim": 1,
                                  "dropout_p": 0, "use_batchnorm": cls_bn},
    }


# ----------------------------------------------------------------------------------------------
# graph topologies
# ----------------------------------------------------------------------------------------------
def cross_camera_edges(cam_sizes, offset=0):
    """Edges as inference.py:209-216 builds them: for each camera, cartesian_prod(nodes in cam, nodes
    in every other cam); nodes are numbered camera by camera.  Row-major => `row` is non-decreasing."""
    starts = np.cumsum([0] + list(cam_sizes))
    n = int(starts[-1])
    rows, cols = [], []
    for c in range(len(cam_sizes)):
        inside = np.arange(starts[c], starts[c + 1])
        outside = np.concatenate([np.arange(0, starts[c]), np.arange(starts[c + 1], n)])
        for i in inside:
            for j in outside:
                rows.append(i + offset)
                cols.append(j + offset)
    return n, np.array([rows, cols], dtype=np.int64).reshape(2, -1)


def dense_edges(n, offset=0):
    """All ordered pairs i != j, i-major."""
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = i != j
    return n, np.stack([i[m] + offset, j[m] + offset]).astype(np.int64)


def union(graphs):
    n_tot, eis = 0, []
    for kind, arg in graphs:
        n, ei = (dense_edges(arg, n_tot) if kind == "dense" else cross_camera_edges(arg, n_tot))
        eis.append(ei)
        n_tot += n
    return n_tot, np.concatenate(eis, axis=1)


# ----------------------------------------------------------------------------------------------
def build_model(MOTMPNet, params, arch, seed, node_mlp_scale):
    torch.manual_seed(seed)
    model = MOTMPNet(copy.deepcopy(params), None, arch)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.weight.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.num_features, generator=g))
        s = torch.tensor(node_mlp_scale, dtype=torch.float32)
        for p in model.MPNet.node_model.node_mlp.parameters():
            p.mul_(s)
    model.eval()
    return model


def make_inputs(n, edge_index, node_in, edge_in, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, node_in, generator=g)
    x = torch.nn.functional.normalize(x, p=2, dim=0)  # mirrors inference.py:189-190 (dim=0)
    edge_attr = torch.rand(edge_index.shape[1], edge_in, generator=g)
    return x.float(), torch.from_numpy(edge_index), edge_attr.float()


class _Data:
    pass


def run_case(MOTMPNet, name, params, arch, n, edge_index, seed_w, seed_in, node_mlp_scale, shared_weights=None,
             edge_perm=None):
    enc = params["encoder_feats_dict"]["nodes"][arch]
    node_in, edge_in = enc["node_in_dim"], params["encoder_feats_dict"]["edges"]["edge_in_dim"]
    model = build_model(MOTMPNet, params, arch, seed_w, node_mlp_scale)
    x, ei, ea = make_inputs(n, edge_index, node_in, edge_in, seed_in)
    if edge_perm is not None:  # same graph, same attributes, edges listed in another order
        ei, ea = ei[:, edge_perm].contiguous(), ea[edge_perm].contiguous()

    trace = {"enc": None, "steps": []}
    h1 = model.encoder.register_forward_hook(lambda m, i, o: trace.__setitem__("enc", o))
    h2 = model.MPNet.register_forward_hook(lambda m, i, o: trace["steps"].append(o))
    data = _Data()
    data.x, data.edge_index, data.edge_attr = x, ei, ea
    with torch.no_grad():
        out = model(data)
    h1.remove()
    h2.remove()

    rec = {
        "params_json": np.array(json.dumps({"model_params": params, "arch": arch})),
        "x": x.numpy(), "edge_index": ei.numpy(), "edge_attr": ea.numpy(),
        "node_mlp_scale": np.float32(node_mlp_scale),
        "n_logits": np.int64(len(out["classified_edges"])),
        "e_enc": trace["enc"][0].numpy(), "h_enc": trace["enc"][1].numpy(),
    }
    for i, t in enumerate(out["classified_edges"]):
        rec[f"logits_{i}"] = t.numpy()
    for i, (h, e) in enumerate(trace["steps"]):
        rec[f"h_step_{i + 1}"] = h.numpy()
        rec[f"e_step_{i + 1}"] = e.numpy()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    if shared_weights is None:
        for k, v in sd.items():
            rec["sd::" + k] = v
    else:
        # Large (node_in=2048) weight sets are stored once; the case keeps the node-MLP tensors (the only
        # ones that differ, through node_mlp_scale) and names the shared file for the rest.
        rec["weights_file"] = np.array(shared_weights)
        for k, v in sd.items():
            if k.startswith("MPNet.node_model"):
                rec["sd::" + k] = v
    np.savez(os.path.join(HERE, name + ".npz"), **rec)
    mx = max(float(t.abs().max()) for t in out["classified_edges"])
    print(f"{name:24s} N={n:4d} E={ei.shape[1]:6d} logits={len(out['classified_edges'])} max|logit|={mx:.3f}")
    return model, sd


def main():
    _install_torch_scatter_standin()
    sys.path.insert(0, REF)
    from models.mpn import MOTMPNet  # the reference, imported unmodified

    tiny = dict(node_in=64, arch="tiny64")

    # --- case 1: N=8 (2 cams x 4), all three aggregators ---------------------------------------
    n, ei = cross_camera_edges([4, 4])
    for agg in ("sum", "mean", "max"):
        run_case(MOTMPNet, f"n8_{agg}", make_params(agg=agg, **tiny), "tiny64", n, ei, 11, 12,
                 0.25 if agg == "sum" else 1.0)

    # --- shared default-config weights (node_in 2048, BN classifier) ----------------------------
    p_def = make_params()
    # case 2: Terrace-shaped 4 cams x 8 = 32 nodes, E = 768
    n, ei = cross_camera_edges([8, 8, 8, 8])
    _, sd = run_case(MOTMPNet, "terrace32", p_def, "resnet50", n, ei, 0, 1, 1.0 / 24, shared_weights="weights_default.npz")
    np.savez(os.path.join(HERE, "weights_default.npz"),
             **{"sd::" + k: v for k, v in sd.items() if not k.startswith("MPNet.node_model")})
    # case 3: dense N=64 (BASELINE config 2) and case 8: the same graph with shuffled edge order
    n, ei = dense_edges(64)
    run_case(MOTMPNet, "dense64", p_def, "resnet50", n, ei, 0, 1, 1.0 / 63, shared_weights="weights_default.npz")

    # --- case 8: shuffled edge order (unsorted `row`); inputs are stored in the shuffled order ---
    # (a small dense graph: the N=64 default-config pair would add another 1.2 MB of fixtures)
    n, ei = dense_edges(24)
    perm = np.random.default_rng(5).permutation(ei.shape[1])
    run_case(MOTMPNet, "dense24_shuffled", make_params(**tiny), "tiny64", n, ei, 21, 22, 1.0 / 23, edge_perm=perm)
    run_case(MOTMPNet, "dense24_sorted", make_params(**tiny), "tiny64", n, ei, 21, 22, 1.0 / 23)

    # --- case 4: disjoint union of 3 graphs of different sizes ------------------------------------
    n, ei = union([("cams", [3, 3]), ("cams", [4, 4, 4]), ("dense", 10)])
    run_case(MOTMPNet, "union3", make_params(**tiny), "tiny64", n, ei, 31, 32, 1.0 / 8)

    # --- case 5: reattach flags -----------------------------------------------------------------
    n, ei = cross_camera_edges([5, 4, 3])
    for rn, re_ in ((True, True), (True, False), (False, True)):
        run_case(MOTMPNet, f"reattach_n{int(rn)}e{int(re_)}",
                 make_params(reattach_nodes=rn, reattach_edges=re_, **tiny), "tiny64", n, ei, 41, 42, 1.0 / 8)

    # --- case 6: classifier BN off (config_training.yaml:181) --------------------------------------
    run_case(MOTMPNet, "bn_off", make_params(cls_bn=False, **tiny), "tiny64", n, ei, 51, 52, 1.0 / 8)

    # --- case 7: number of steps -----------------------------------------------------------------
    for L in (0, 1, 2, 8):
        run_case(MOTMPNet, f"steps_L{L}", make_params(L=L, **tiny), "tiny64", n, ei, 61, 62, 1.0 / 8)
    run_case(MOTMPNet, "steps_L4_c1", make_params(L=4, n_cls=1, **tiny), "tiny64", n, ei, 61, 62, 1.0 / 8)

    # --- case 9: arch='bdnet_market' (node_in 512) --------------------------------------------------
    n, ei = cross_camera_edges([6, 5, 5])
    run_case(MOTMPNet, "bdnet512", make_params(node_in=512, arch="bdnet_market"), "bdnet_market", n, ei, 71, 72, 1.0 / 10)

    # --- extra: ONLY_APPEARANCE-style 2-d edge attributes (inference.py:246-247) -------------------
    run_case(MOTMPNet, "edge_in2", make_params(edge_in=2, **tiny), "tiny64", n, ei, 81, 82, 1.0 / 10)

    # --- extra: nodes without out-edges / isolated nodes, arbitrary (unsorted) directed graph --------
    rng = np.random.default_rng(7)
    n = 20
    rows = rng.integers(0, 14, size=90)  # nodes 14..19 never appear as a source
    cols = rng.integers(0, 18, size=90)  # nodes 18,19 are isolated
    ei = np.stack([rows, cols]).astype(np.int64)
    for agg in ("sum", "mean", "max"):
        run_case(MOTMPNet, f"ragged_{agg}", make_params(agg=agg, **tiny), "tiny64", n, ei, 91, 92, 0.2 if agg == "sum" else 1.0)

    # --- extra: generic widths, multi-layer MLPs, BN + dropout modules inside encoder/MPN (eval) -----
    n, ei = cross_camera_edges([4, 3, 3])
    run_case(MOTMPNet, "generic_dims",
             make_params(node_in=40, node_fc=(48, 24), node_out=16, edge_in=3, edge_fc=(5,), edge_out=8,
                         edge_mlp_fc=(10, 8), node_mlp_fc=(24, 16), cls_fc=(5, 3), cls_bn=True, enc_bn=True,
                         enc_dropout=0.4, mpn_bn=True, mpn_dropout=0.3, arch="generic"),
             "generic", n, ei, 101, 102, 1.0 / 6)

    # --- extra: generic family with both reattach flags, 'max', single-layer encoder, bare Linear classifier, on
    #     an unsorted ragged graph; and the default widths with a three-layer classifier + 'mean' -------------------
    rng = np.random.default_rng(17)
    n = 18
    ei = np.stack([rng.integers(0, 15, size=70), rng.integers(0, 18, size=70)]).astype(np.int64)
    run_case(MOTMPNet, "generic_reattach_max",
             make_params(node_in=24, node_fc=(), node_out=12, edge_in=3, edge_fc=(), edge_out=5, edge_mlp_fc=(7, 5),
                         node_mlp_fc=(12,), cls_fc=(), cls_bn=False, agg="max", reattach_nodes=True, reattach_edges=True,
                         arch="generic"), "generic", n, ei, 111, 112, 1.0)
    n, ei = cross_camera_edges([5, 4, 4])
    run_case(MOTMPNet, "generic_deepcls_mean",
             make_params(node_in=64, cls_fc=(8, 4), agg="mean", arch="tiny64"), "tiny64", n, ei, 121, 122, 1.0)


if __name__ == "__main__":
    main()

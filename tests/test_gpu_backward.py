"""Row N3 on the GPU: loss.backward() through gnn_cca_amd.MOTMPNet in train mode against the gradients the REFERENCE's
own module produces under torch autograd (tests/golden/bwd_*.npz), and against the autograd oracle at a larger size."""
import copy
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle.mpn_oracle import TorchTrainOracle
from test_backward_oracle import DROP_CASES, load_bwd

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "bwd_*.npz")))


class Data:
    def __init__(self, x, edge_index, edge_attr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr


def build(params, arch, sd):
    from gnn_cca_amd import MOTMPNet
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.cuda().train()


def loss_of(out, labels):
    crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
    return sum(crit(t.view(-1), labels) for t in out["classified_edges"])  # train.py:80-97


@pytest.mark.parametrize("name", CASES)
def test_gradients_match_reference(name):
    params, arch, sd, grads, _, a = load_bwd(name)
    m = build(params, arch, sd)
    d = Data(torch.from_numpy(a["x"]).cuda(), torch.from_numpy(a["edge_index"]).cuda(), torch.from_numpy(a["edge_attr"]).cuda())
    out = m(d)
    loss = loss_of(out, torch.from_numpy(a["labels"]).cuda())
    loss.backward()
    assert abs(float(loss) - float(a["loss"])) <= 5e-6
    for i, t in enumerate(out["classified_edges"]):
        assert np.abs(t.detach().cpu().numpy() - a[f"logits_{i}"]).max() <= 5e-6
    got = dict(m.named_parameters())
    assert sorted(got) == sorted(grads)
    for k, ref in grads.items():
        g = got[k].grad
        assert g is not None, k
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(g.cpu().numpy() - ref).max() <= 2e-5 * scale, (k, float(np.abs(g.cpu().numpy() - ref).max()))
    _, _, _, _, after, _ = load_bwd(name)
    state = m.state_dict()
    for k, v in after.items():  # BatchNorm buffers after the train-mode forward (cls_bn_train)
        assert np.abs(state[k].cpu().numpy().astype(np.float64) - v).max() <= 1e-6, k


def test_gradients_dense64_default_width_vs_oracle():
    """node_in 2048 / BASELINE config-2 graph, classifier without BN: the training config's shape end to end."""
    from oracle.mpn_oracle import load_case
    params, arch, sd, a = load_case(os.path.join(GOLDEN_DIR, "dense64.npz"))
    params = copy.deepcopy(params)
    params["classifier_feats_dict"]["use_batchnorm"] = False
    sd = {k: v for k, v in sd.items() if ".fc_layers.1." not in k or not k.startswith("classifier")}
    sd["classifier.edge_mlp.fc_layers.3.weight"] = sd.pop("classifier.edge_mlp.fc_layers.4.weight")
    sd["classifier.edge_mlp.fc_layers.3.bias"] = sd.pop("classifier.edge_mlp.fc_layers.4.bias")
    labels = (np.random.default_rng(0).random(a["edge_index"].shape[1]) < 0.2).astype(np.float32)
    ref_loss, _, ref = TorchTrainOracle(params, arch, sd).loss_and_grads(a["x"], a["edge_index"], a["edge_attr"], labels)
    m = build(params, arch, sd)
    out = m(Data(torch.from_numpy(a["x"]).cuda(), torch.from_numpy(a["edge_index"]).cuda(), torch.from_numpy(a["edge_attr"]).cuda()))
    loss = loss_of(out, torch.from_numpy(labels).cuda())
    loss.backward()
    assert abs(float(loss) - ref_loss) <= 5e-6
    for k, p in m.named_parameters():
        r = ref[k].numpy()
        scale = max(1.0, float(np.abs(r).max()))
        assert np.abs(p.grad.cpu().numpy() - r).max() <= 3e-5 * scale, k


def test_sgd_steps_track_the_oracle():
    """Three optimizer steps: weights are re-packed after every in-place update (the HBM blob cache keys on versions)."""
    params, arch, sd, _, _, a = load_bwd("terrace32")
    m = build(params, arch, sd)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    d = Data(torch.from_numpy(a["x"]).cuda(), torch.from_numpy(a["edge_index"]).cuda(), torch.from_numpy(a["edge_attr"]).cuda())
    lab = torch.from_numpy(a["labels"]).cuda()
    cur = {k: np.asarray(v) for k, v in sd.items()}
    losses = []
    for _ in range(3):
        ref_loss, _, ref = TorchTrainOracle(params, arch, cur).loss_and_grads(a["x"], a["edge_index"], a["edge_attr"], a["labels"])
        opt.zero_grad()
        loss = loss_of(m(d), lab)
        loss.backward()
        opt.step()
        assert abs(float(loss) - ref_loss) <= 1e-5
        cur = {k: (cur[k] - 0.05 * ref[k].numpy() if k in ref else cur[k]) for k in cur}
        losses.append(float(loss))
    assert losses[2] < losses[0]
    m.eval()
    with torch.no_grad():
        m(d)  # eval after training: the blob is rebuilt from the updated parameters


@pytest.mark.parametrize("agg,re_n,re_e", [("sum", True, True), ("max", True, False), ("mean", False, True)])
def test_random_graphs_reattach_gradients_vs_oracle(agg, re_n, re_e):
    """The reattach_initial_nodes / reattach_initial_edges variants (models/mpn.py:283-285) on irregular graphs: randomly
    initialised weights of the widened shapes, every parameter gradient against torch autograd over the CPU oracle."""
    from gnn_cca_amd import MOTMPNet
    from oracle.mpn_oracle import load_case
    from test_gpu_fuzz import random_graph
    params, arch, _, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))
    params = copy.deepcopy(params)
    params.update(node_agg_fn=agg, reattach_initial_nodes=re_n, reattach_initial_edges=re_e)
    params["classifier_feats_dict"]["use_batchnorm"] = False
    torch.manual_seed(11)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    with torch.no_grad():
        for prm in m.MPNet.node_model.node_mlp.parameters():
            prm.mul_(0.25 if agg == "sum" else 1.0)
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    orc = TorchTrainOracle(params, arch, sd)
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((agg, re_n, re_e)).encode()))
    for it in range(16):
        kind = ["chunks", "sparse", "frames", "frames"][it % 4]
        n, ei = random_graph(rng, kind)
        if it % 4 == 3:
            ei = ei[:, rng.permutation(ei.shape[1])]  # unsorted rows
        while ei.shape[1] < 2:
            n, ei = random_graph(rng, "frames")
        x = (rng.standard_normal((n, 64)) * 0.5).astype(np.float32)
        ea = rng.random((ei.shape[1], 4)).astype(np.float32)
        labels = (rng.random(ei.shape[1]) < 0.3).astype(np.float32)
        ref_loss, _, ref = orc.loss_and_grads(x, ei, ea, labels)
        m.zero_grad(set_to_none=True)
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))
        loss = loss_of(out, torch.from_numpy(labels).cuda())
        loss.backward()
        assert abs(float(loss) - ref_loss) <= 1e-5, (it, float(loss), ref_loss)
        for k, prm in m.named_parameters():
            r = ref[k].numpy()
            scale = max(1.0, float(np.abs(r).max()))
            err = float(np.abs(prm.grad.cpu().numpy() - r).max())
            assert err <= 3e-5 * scale, (it, n, ei.shape, k, err)


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_random_graphs_gradients_vs_oracle(agg):
    """Irregular inputs for the backward kernels: hubs whose degree straddles the 64-edge chunks and the 256-edge
    workgroup chunk, isolated nodes, duplicate edges, self loops and UNSORTED rows (row-indexed gradients then leave the
    LDS pre-accumulation window and go straight to global memory) -- every parameter gradient against torch autograd
    over the CPU oracle."""
    from oracle.mpn_oracle import load_case
    from test_gpu_fuzz import random_graph
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))  # node_in 64, BatchNorm inside the classifier
    params = copy.deepcopy(params)
    params.update(node_agg_fn=agg)
    sd = dict(sd)
    if agg != "sum":
        for k in list(sd):
            if k.startswith("MPNet.node_model"):
                sd[k] = (sd[k] * np.float32(4.0)).astype(np.float32)
    m = build(params, arch, sd)
    orc = TorchTrainOracle(params, arch, sd)
    import zlib
    rng = np.random.default_rng(zlib.crc32(agg.encode()))
    for it in range(24):
        kind = ["chunks", "sparse", "unsorted", "frames"][it % 4]
        if agg == "max" and kind == "unsorted":
            # dozens of parallel edges between the same few nodes make the arg-max ill-conditioned (two messages one ulp
            # apart: the matmul-based oracle and the FMA-chain kernels may crown different edges); unsorted rows are
            # covered for 'max' by a shuffled cross-camera graph instead
            n, ei = random_graph(rng, "frames")
            ei = ei[:, rng.permutation(ei.shape[1])]
        else:
            n, ei = random_graph(rng, kind)
        while ei.shape[1] < 2:  # train-mode BatchNorm1d (classifier) refuses a batch of one edge, in torch as here
            n, ei = random_graph(rng, "frames")
        x = (rng.standard_normal((n, 64)) * 0.3).astype(np.float32)
        ea = rng.random((ei.shape[1], 4)).astype(np.float32)
        labels = (rng.random(ei.shape[1]) < 0.3).astype(np.float32)
        ref_loss, _, ref = orc.loss_and_grads(x, ei, ea, labels)
        m.zero_grad(set_to_none=True)
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))
        loss = loss_of(out, torch.from_numpy(labels).cuda())
        loss.backward()
        assert abs(float(loss) - ref_loss) <= 1e-5, (it, float(loss), ref_loss)
        for k, p in m.named_parameters():
            r = ref[k].numpy()
            scale = max(1.0, float(np.abs(r).max()))
            err = float(np.abs(p.grad.cpu().numpy() - r).max())
            # The bias of a Linear that feeds a train-mode BatchNorm has an analytically ZERO gradient (the batch mean
            # absorbs it): both sides hold only the rounding residue of ~1e-2-sized terms cancelling, 1e-5 in size.
            if k == "classifier.edge_mlp.fc_layers.0.bias":
                assert float(np.abs(p.grad.cpu().numpy()).max()) <= 5e-4 and float(np.abs(r).max()) <= 5e-4, (it, k)
                continue
            assert err <= 3e-5 * scale, (it, n, ei.shape, k, err)


@pytest.mark.parametrize("name", ["terrace32", "terrace32_max", "terrace32_mean"])
@pytest.mark.parametrize("which,bad", [(0, 10 ** 6), (1, -3), (0, 2 ** 40 + 1)])
def test_out_of_range_index_in_train_mode_is_contained(name, which, bad):
    """An index outside [0, N) in train mode (the reference raises IndexError at models/mpn.py:48): the forward flags it
    and poisons the logits; the backward must treat the edge as dead -- no read, write or atomic at a wild address.  A
    canary tensor allocated right around the call stays intact, the flag word says BAD_INDEX, loss and gradients are
    NaN (never silently plausible numbers)."""
    params, arch, sd, _, _, a = load_bwd(name)
    m = build(params, arch, sd)
    ei = torch.from_numpy(a["edge_index"]).clone()
    ei[which, ei.shape[1] // 2] = bad
    canary_lo = torch.full((1 << 18,), 7.0, device="cuda")
    d = Data(torch.from_numpy(a["x"]).cuda(), ei.cuda(), torch.from_numpy(a["edge_attr"]).cuda())
    canary_hi = torch.full((1 << 18,), 7.0, device="cuda")
    out = m(d)
    loss = loss_of(out, torch.from_numpy(a["labels"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert m.graph_flags() & 2
    assert not np.isfinite(float(loss))
    assert all(torch.isnan(t).all().item() for t in out["classified_edges"])
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
    assert bool((canary_lo == 7.0).all()) and bool((canary_hi == 7.0).all())


def test_max_aggregation_gradient_goes_to_the_first_arg_max():
    """torch_scatter's scatter_max hands the whole gradient to its `arg`, the FIRST edge that attains a node's maximum
    (torch's own amax backward would split it among tied edges).  Constructed exact ties that make the two rules differ in
    a PARAMETER gradient: eight channels of the node MLP get a zero edge-feature block, so every edge of a node has the
    same message Q[row][c] in those channels while its edge features differ -- d W_ne[c] is gh * e'(first edge) under the
    reference's rule and gh * mean(e') under the other.  Checked against autograd over the oracle's first-arg scatter_max."""
    from oracle.mpn_oracle import load_case
    from test_gpu_fuzz import random_graph
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "n8_sum.npz"))
    params = copy.deepcopy(params)
    params.update(node_agg_fn="max")
    params["classifier_feats_dict"]["use_batchnorm"] = False
    sd = {k: np.array(v) for k, v in sd.items() if "classifier" not in k}
    torch.manual_seed(5)
    from gnn_cca_amd import MOTMPNet
    fresh = MOTMPNet(copy.deepcopy(params), None, arch).state_dict()
    for k, v in fresh.items():
        if k not in sd:
            sd[k] = v.numpy().copy()
    w = sd["MPNet.node_model.node_mlp.fc_layers.0.weight"] * np.float32(4.0)
    b = sd["MPNet.node_model.node_mlp.fc_layers.0.bias"] * np.float32(4.0)
    w[:8, 32:] = 0.0                    # channels 0..7 ignore the edge features: exact ties among all edges of a node
    b[:8] = np.abs(b[:8]) + 0.05        # ... at a positive value, so the tie carries gradient
    sd["MPNet.node_model.node_mlp.fc_layers.0.weight"], sd["MPNet.node_model.node_mlp.fc_layers.0.bias"] = w, b
    m = build(params, arch, sd)
    orc = TorchTrainOracle(params, arch, sd)
    rng = np.random.default_rng(77)
    for it in range(6):
        n, ei = random_graph(rng, "frames")
        if it % 2:
            ei = ei[:, rng.permutation(ei.shape[1])]          # the first edge by ID, not by sorted position
        x = (rng.standard_normal((n, 64)) * 0.3).astype(np.float32)
        ea = rng.random((ei.shape[1], 4)).astype(np.float32)
        labels = (rng.random(ei.shape[1]) < 0.3).astype(np.float32)
        ref_loss, _, ref = orc.loss_and_grads(x, ei, ea, labels)
        m.zero_grad(set_to_none=True)
        out = m(Data(torch.from_numpy(x).cuda(), torch.from_numpy(ei).cuda(), torch.from_numpy(ea).cuda()))
        loss = loss_of(out, torch.from_numpy(labels).cuda())
        loss.backward()
        assert abs(float(loss) - ref_loss) <= 1e-5
        gw = m.MPNet.node_model.node_mlp.fc_layers[0].weight.grad.cpu().numpy()
        rw = ref["MPNet.node_model.node_mlp.fc_layers.0.weight"].numpy()
        assert float(np.abs(rw[:8, 32:]).max()) > 1e-4, "the constructed ties must carry gradient"
        for k, p in m.named_parameters():
            r = ref[k].numpy()
            scale = max(1.0, float(np.abs(r).max()))
            assert float(np.abs(p.grad.cpu().numpy() - r).max()) <= 3e-5 * scale, (it, k)
        assert float(np.abs(gw[:8, 32:] - rw[:8, 32:]).max()) <= 3e-5


def _dropout_params(name, p_enc, p_edge, p_node, p_cls, **over):
    """GRAPH_NET_PARAMS of a backward golden with Dropout switched on in the four MLP groups (models/mlp.py:20-21)."""
    params, arch, sd, _, _, a = load_bwd(name)
    params = copy.deepcopy(params)
    params.update(over)
    params["encoder_feats_dict"]["nodes"][arch]["dropout_p"] = p_enc
    params["edge_model_feats_dict"]["dropout_p"] = p_edge
    params["node_model_feats_dict"]["dropout_p"] = p_node
    params["classifier_feats_dict"]["dropout_p"] = p_cls
    return params, arch, sd, a


@pytest.mark.parametrize("name,ps,over", [
    ("terrace32", (0.2, 0.1, 0.3, 0.25), {}),
    ("terrace32", (0.0, 0.0, 0.5, 0.0), {}),                 # only the messages
    ("terrace32_mean", (0.3, 0.2, 0.0, 0.1), {}),
    ("terrace32_max", (0.1, 0.1, 0.4, 0.1), {}),             # the maximum is taken over the messages AFTER Dropout
    ("terrace32_reatt_ne_mean", (0.2, 0.2, 0.2, 0.2), {}),
    ("cls_bn_train", (0.15, 0.1, 0.2, 0.3), {}),             # train-mode BatchNorm -> ReLU -> Dropout in the classifier
    ("dense20_shuf", (0.2, 0.1, 0.3, 0.2), {}),              # unsorted rows: masks are indexed by the CALLER's edge ids
])
def test_train_mode_dropout_matches_autograd_oracle(name, ps, over):
    """Dropout with p > 0 in train mode (row N3): logits, loss and every parameter gradient against torch autograd over the
    CPU oracle, which applies the SAME masks (oracle.dropout_scale is the numpy twin of the kernels' counter-based hash of
    (seed, tensor, element); nothing is stored between forward and backward on either side)."""
    from gnn_cca_amd import MOTMPNet
    params, arch, sd, a = _dropout_params(name, *ps, **over)
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().train()
    seed = 0x1234ABCD5678
    m.set_dropout_seed(seed)
    orc = TorchTrainOracle(params, arch, sd, dropout=dict(p_enc=ps[0], p_edge=ps[1], p_node=ps[2], p_cls=ps[3], seed=seed))
    ref_loss, ref_logits, ref = orc.loss_and_grads(a["x"], a["edge_index"], a["edge_attr"], a["labels"])
    d = Data(torch.from_numpy(a["x"]).cuda(), torch.from_numpy(a["edge_index"]).cuda(), torch.from_numpy(a["edge_attr"]).cuda())
    out = m(d)
    loss = loss_of(out, torch.from_numpy(a["labels"]).cuda())
    loss.backward()
    for t, r in zip(out["classified_edges"], ref_logits):
        assert np.abs(t.detach().cpu().numpy() - r.numpy()).max() <= 2e-5, name
    assert abs(float(loss) - ref_loss) <= 1e-5
    for k, prm in m.named_parameters():
        r = ref[k].numpy()
        scale = max(1.0, float(np.abs(r).max()))
        if k == "classifier.edge_mlp.fc_layers.0.bias" and name == "cls_bn_train":
            continue   # analytically zero behind a train-mode BatchNorm: rounding residue on both sides
        assert float(np.abs(prm.grad.cpu().numpy() - r).max()) <= 5e-5 * scale, (name, k)
    # the seed word advanced: the next forward draws different masks; re-seeding reproduces the first one bit for bit
    first = torch.cat([t.detach().view(-1) for t in out["classified_edges"]])
    with torch.no_grad():
        again = torch.cat([t.view(-1) for t in m(d)["classified_edges"]])
        m.set_dropout_seed(seed)
        same = torch.cat([t.view(-1) for t in m(d)["classified_edges"]])
    if name == "terrace32":   # (some goldens have constant logits: their dead ReLUs hide the masks)
        assert not torch.equal(again, first)
    assert torch.equal(same, first)
    m.eval()
    with torch.no_grad():   # eval: Dropout is the identity again
        ev = m(d)["classified_edges"][-1]
    assert torch.isfinite(ev).all()


def test_dropout_keep_rate_and_expectation():
    """Statistics of the device masks: with only the message Dropout on, the keep rate is 1 - p and E[logit] over seeds
    approaches the eval-mode logit's neighbourhood (inverted-dropout scaling 1 / (1 - p))."""
    from oracle.mpn_oracle import dropout_scale
    keep = np.mean([float((dropout_scale(s, 48 + 1, 768, 32, 0.4) > 0).mean()) for s in range(20)])
    assert abs(keep - 0.6) < 0.01
    sc = dropout_scale(7, 3, 100, 6, 0.25)
    assert set(np.unique(sc).tolist()) <= {0.0, np.float32(1.0 / 0.75)}


@pytest.mark.parametrize("name", DROP_CASES)
def test_train_mode_dropout_matches_reference_golden(name):
    """The HIP path in train mode with Dropout against the REFERENCE's own module run with the same masks injected
    (tests/golden/drop_*.npz, make_golden_dropout.py): logits, loss, every parameter gradient."""
    from gnn_cca_amd import MOTMPNet
    params, arch, sd, grads, _, a = load_bwd(name, "drop_")
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().train()
    m.set_dropout_seed(int(a["dropout_seed"]))
    assert [round(q, 6) for q in m._dropout_ps()] == [round(float(v), 6) for v in a["dropout_p"]]
    d = Data(torch.from_numpy(a["x"]).cuda(), torch.from_numpy(a["edge_index"]).cuda(), torch.from_numpy(a["edge_attr"]).cuda())
    out = m(d)
    loss = loss_of(out, torch.from_numpy(a["labels"]).cuda())
    loss.backward()
    assert abs(float(loss) - float(a["loss"])) <= 1e-5
    for i, t in enumerate(out["classified_edges"]):
        assert np.abs(t.detach().cpu().numpy() - a[f"logits_{i}"]).max() <= 2e-5
    for k, ref in grads.items():
        g = dict(m.named_parameters())[k].grad
        scale = max(1.0, float(np.abs(ref).max()))
        if k == "classifier.edge_mlp.fc_layers.0.bias" and "cls_bn" in name:
            continue   # analytically zero behind a train-mode BatchNorm
        assert np.abs(g.cpu().numpy() - ref).max() <= 5e-5 * scale, (name, k)

"""Forward hooks on encoder / MPNet / classifier in TRAIN mode (missing 3): the reference fires them in both modes
(models/mpn.py:270,288,292 are ordinary nn.Module calls).  MOTMPNet.forward replays the container calls from the latents the
training forward saved for its backward -- the fused engine's trace, the layer-by-layer engine's tape -- so hooks see TRAIN-mode
values (BatchNorm with batch statistics, Dropout applied), the classifier outputs are the autograd-connected logits, and the
gradients are what they are without hooks.  Checked against the autograd oracle's train-mode latents (pinned by the reference's own
logits / gradients in tests/golden/bwd_*.npz and lw_*.npz)."""
import numpy as np
import pytest
import torch

from oracle.mpn_oracle import TorchTrainOracle
from test_backward_oracle import load_bwd
from test_gpu_train_layerwise import build, data_of, loss_of

pytestmark = pytest.mark.gpu


class TapOracle(TorchTrainOracle):
    """TorchTrainOracle that also records what the containers return: encoder outputs, (h, e) after every MPNet call."""

    def forward(self, x, edge_index, edge_attr):
        self.taps = {"e_steps": [], "h_steps": []}
        mlp, agg = self._mlp, self._aggregate

        def mlp_tap(prefix, xx, *a, **k):
            out = mlp(prefix, xx, *a, **k)
            if prefix == "encoder.edge_mlp":
                self.taps["e_enc"] = out.detach()
            elif prefix == "encoder.node_mlp":
                self.taps["h_enc"] = out.detach()
            elif prefix == "MPNet.edge_model.edge_mlp":
                self.taps["e_steps"].append(out.detach())
            return out

        def agg_tap(flow, row, n):
            out = agg(flow, row, n)
            self.taps["h_steps"].append(out.detach())
            return out

        self._mlp, self._aggregate = mlp_tap, agg_tap
        try:
            return super().forward(x, edge_index, edge_attr)
        finally:
            self._mlp, self._aggregate = mlp, agg


CASES = [("bwd_", "terrace32", "auto"), ("bwd_", "cls_bn_train", "auto"), ("bwd_", "terrace32_reatt_ne_mean", "auto"),
         ("bwd_", "terrace32", "layerwise"), ("lw_", "bn_everywhere", "auto"), ("lw_", "generic_dims", "auto")]


@pytest.mark.parametrize("prefix,name,engine", CASES)
def test_train_mode_hooks_fire_with_train_mode_latents(prefix, name, engine):
    params, arch, sd, grads, after, a = load_bwd(name, prefix)
    labels = torch.from_numpy(np.asarray(a["labels"])).cuda().float()
    # Dropout (the generic golden has it): both models and the oracle draw the golden's masks
    seed = int(a["dropout_seed"]) if "dropout_seed" in a else None
    ps = [float(v) for v in a["dropout_p"]] if "dropout_p" in a else [0.0] * 4
    # without hooks: the reference run
    m0 = build(params, arch, sd, engine)
    if seed is not None:
        m0.set_dropout_seed(seed)
    out0 = m0(data_of(a))
    loss_of(out0, labels).backward()
    # with hooks
    m = build(params, arch, sd, engine)
    if seed is not None:
        m.set_dropout_seed(seed)
    d = data_of(a)
    seen = {"enc": [], "mp_in": [], "mp_out": [], "cls": []}
    hooks = [m.encoder.register_forward_hook(lambda mod, i, o: seen["enc"].append(o)),
             m.MPNet.register_forward_pre_hook(lambda mod, i: seen["mp_in"].append(i)),
             m.MPNet.register_forward_hook(lambda mod, i, o: seen["mp_out"].append(o)),
             m.classifier.register_forward_hook(lambda mod, i, o: seen["cls"].append(o))]
    out = m(d)
    loss = loss_of(out, labels)
    loss.backward()
    for h in hooks:
        h.remove()
    L = int(params["num_enc_steps"])
    assert len(seen["enc"]) == 1 and len(seen["mp_out"]) == L and len(seen["cls"]) == len(out["classified_edges"])
    # same logits as without hooks, bit for bit (the same kernels ran); same gradients up to the order of the backward's atomic sums
    for s, t in zip(out["classified_edges"], out0["classified_edges"]):
        assert s.requires_grad and torch.equal(s.detach(), t.detach())
    for (k, p), (_, p0) in zip(m.named_parameters(), m0.named_parameters()):
        assert p.grad is not None and torch.allclose(p.grad, p0.grad, rtol=1e-4, atol=1e-6 * max(1.0, float(p0.grad.abs().max()))), k
    for (dec, none), o in zip(seen["cls"], out["classified_edges"]):
        assert none is None and dec is o
    # the latents are the train-mode ones of the autograd oracle
    orc = TapOracle(params, arch, sd, dropout=(dict(p_enc=ps[0], p_edge=ps[1], p_node=ps[2], p_cls=ps[3], seed=seed)
                                               if seed is not None and any(q > 0 for q in ps) else None))
    orc.forward(a["x"], a["edge_index"], a["edge_attr"])
    e_enc, h_enc = seen["enc"][0]                      # (edge_out, node_out): edge first (models/mpn.py:142)
    tol = 2e-5
    assert np.abs(e_enc.cpu().numpy() - orc.taps["e_enc"].numpy()).max() <= tol
    assert np.abs(h_enc.cpu().numpy() - orc.taps["h_enc"].numpy()).max() <= tol
    nf = 2 if params["reattach_initial_nodes"] else 1
    ef = 2 if params["reattach_initial_edges"] else 1
    for s in range(L):
        x_in, ei_in, e_in = seen["mp_in"][s]
        assert x_in.shape[1] == nf * h_enc.shape[1] and e_in.shape[1] == ef * e_enc.shape[1] and ei_in is d.edge_index
        h_s, e_s = seen["mp_out"][s]
        assert np.abs(h_s.cpu().numpy() - orc.taps["h_steps"][s].numpy()).max() <= tol * max(1.0, float(orc.taps["h_steps"][s].abs().max())), s
        assert np.abs(e_s.cpu().numpy() - orc.taps["e_steps"][s].numpy()).max() <= tol * max(1.0, float(orc.taps["e_steps"][s].abs().max())), s
    # un-hooked again: no replay, nothing recorded
    n_before = len(seen["enc"])
    m.zero_grad()
    m(d)
    assert len(seen["enc"]) == n_before

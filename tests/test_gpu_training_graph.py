"""GraphedTrainStep: the whole training iteration replayed from one HIP graph trains exactly like the eager loop."""
import copy

import numpy as np
import pytest
import torch

from test_backward_oracle import load_bwd

pytestmark = pytest.mark.gpu


class Data:
    pass


def _setup(seed=0):
    from gnn_cca_amd import MOTMPNet
    params, arch, sd, _, _, a = load_bwd("terrace32")
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.cuda().train()
    d = Data()
    d.x, d.edge_index, d.edge_attr = (torch.from_numpy(a[k]).cuda() for k in ("x", "edge_index", "edge_attr"))
    labels = torch.from_numpy(np.asarray(a["labels"])).cuda().float()
    return m, d, labels


def test_graphed_step_trains_like_eager():
    from gnn_cca_amd.training import GraphedTrainStep
    crit = torch.nn.BCEWithLogitsLoss()
    loss_fn = lambda out, lab: sum(crit(t.view(-1), lab) for t in out["classified_edges"])
    m1, d, labels = _setup()
    m2, _, _ = _setup()
    o1 = torch.optim.SGD(m1.parameters(), lr=0.05)
    o2 = torch.optim.SGD(m2.parameters(), lr=0.05)
    step = GraphedTrainStep(m2, o2, loss_fn, warmup=2)
    rng = np.random.default_rng(0)
    losses1, losses2 = [], []
    for it in range(7):  # 2 eager warm-ups, 1 captured, 4 replayed -- with a different batch every time
        d.edge_attr = torch.from_numpy(rng.random(tuple(d.edge_attr.shape)).astype(np.float32)).cuda()
        o1.zero_grad()
        l1 = loss_fn(m1(d), labels)
        l1.backward()
        o1.step()
        losses1.append(float(l1))
        losses2.append(float(step(d, labels)))
    assert len(step._graphs) == 1
    assert np.allclose(losses1, losses2, rtol=2e-5, atol=1e-6), (losses1, losses2)
    assert losses1[-1] < losses1[0]
    for (k, p1), (_, p2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.allclose(p1.float(), p2.float(), rtol=1e-4, atol=1e-6), k


def test_graphed_step_with_dropout_draws_fresh_masks_every_replay():
    """Dropout under whole-step capture: the seed lives in a device word that the captured step copies and advances, so every
    replay draws new masks -- the same sequence of masks, hence of losses and weights, as the eager loop from the same seed."""
    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.training import GraphedTrainStep
    params, arch, sd, _, _, a = load_bwd("terrace32")
    params = copy.deepcopy(params)
    params["encoder_feats_dict"]["nodes"][arch]["dropout_p"] = 0.2
    params["edge_model_feats_dict"]["dropout_p"] = 0.1
    params["node_model_feats_dict"]["dropout_p"] = 0.3
    params["classifier_feats_dict"]["dropout_p"] = 0.2
    crit = torch.nn.BCEWithLogitsLoss()
    loss_fn = lambda out, lab: sum(crit(t.view(-1), lab) for t in out["classified_edges"])
    models = []
    for _ in range(2):
        m = MOTMPNet(copy.deepcopy(params), None, arch)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.cuda().train()
        m.set_dropout_seed(424242)
        models.append(m)
    m1, m2 = models
    d = Data()
    d.x, d.edge_index, d.edge_attr = (torch.from_numpy(a[k]).cuda() for k in ("x", "edge_index", "edge_attr"))
    labels = torch.from_numpy(np.asarray(a["labels"])).cuda().float()
    o1 = torch.optim.SGD(m1.parameters(), lr=0.05)
    o2 = torch.optim.SGD(m2.parameters(), lr=0.05)
    step = GraphedTrainStep(m2, o2, loss_fn, warmup=2)
    losses1, losses2 = [], []
    for it in range(7):
        o1.zero_grad()
        l1 = loss_fn(m1(d), labels)
        l1.backward()
        o1.step()
        losses1.append(float(l1))
        losses2.append(float(step(d, labels)))
    assert len(step._graphs) == 1
    assert np.allclose(losses1, losses2, rtol=2e-5, atol=1e-6), (losses1, losses2)
    assert len({round(v, 6) for v in losses2[3:]}) > 1   # replays do not repeat one mask


def test_graphed_step_on_the_layerwise_engine_with_batchnorm_and_dropout():
    """The layer-by-layer engine (BatchNorm in every MLP + Dropout: csrc/train_generic.cuh) under whole-step capture: its memsets,
    copies, one-launch-per-op kernels, BatchNorm buffer updates and num_batches_tracked increments all replay; losses and final
    state equal the eager loop's from the same seed."""
    from gnn_cca_amd import MOTMPNet
    from gnn_cca_amd.training import GraphedTrainStep
    params, arch, sd, _, _, a = load_bwd("bn_drop_mean", "lw_")
    crit = torch.nn.BCEWithLogitsLoss()
    loss_fn = lambda out, lab: sum(crit(t.view(-1), lab) for t in out["classified_edges"])
    models = []
    for _ in range(2):
        m = MOTMPNet(copy.deepcopy(params), None, arch)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m = m.cuda().train()
        m.set_dropout_seed(777)
        models.append(m)
    m1, m2 = models
    d = Data()
    d.x, d.edge_index, d.edge_attr = (torch.from_numpy(a[k]).cuda() for k in ("x", "edge_index", "edge_attr"))
    labels = torch.from_numpy(np.asarray(a["labels"])).cuda().float()
    o1 = torch.optim.SGD(m1.parameters(), lr=0.02)
    o2 = torch.optim.SGD(m2.parameters(), lr=0.02)
    step = GraphedTrainStep(m2, o2, loss_fn, warmup=2)
    losses1, losses2 = [], []
    for it in range(6):
        o1.zero_grad()
        l1 = loss_fn(m1(d), labels)
        l1.backward()
        o1.step()
        losses1.append(float(l1))
        losses2.append(float(step(d, labels)))
    assert m1._train_path == "layerwise" and m2._train_path == "layerwise"
    assert len(step._graphs) == 1
    # (the engine's scatter-adds are float atomics: eager and replayed steps differ by summation order, which BatchNorm + Dropout
    # amplify a little over six steps)
    assert np.allclose(losses1, losses2, rtol=1e-3, atol=1e-5), (losses1, losses2)
    s1, s2 = m1.state_dict(), m2.state_dict()
    for k in s1:
        if "num_batches_tracked" in k:
            assert int(s1[k]) == int(s2[k]) > 0, k
        else:
            assert torch.allclose(s1[k].float(), s2[k].float(), rtol=1e-2, atol=1e-4), k

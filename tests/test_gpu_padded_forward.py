"""gnn_cca_amd.inference.PaddedForward: every frame of the per-frame loop (inference.py:173-283) padded to ONE shape by
`gnncca_pad_frame` and served by ONE HIP graph.  The padding is a component of its own (dummy nodes with self loops), so the
frame's own logits must be those of the frame alone: checked against the eager forward of the unpadded frame (to rounding: the
encoder's reduction order follows the row count) and against the CPU oracle on the unpadded frame."""
import numpy as np
import pytest
import torch

from oracle.mpn_oracle import NumpyOracle  # the checker
from test_gpu_column_ranges import cross_camera_graph, inputs
from test_gpu_inference_graph import Data, _model

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _frames(seed=3):
    """Reference-shaped frames (inference.py:209-216) of many shapes, one with an empty camera, one with a lone detection per camera."""
    rng = np.random.default_rng(seed)
    cams = [[8, 8, 8, 8], [3, 0, 5], [1, 1, 1, 1], [9, 7, 6, 8], [2, 11, 4, 3], [5, 5], [10, 10, 10, 4]]
    out = []
    for k, c in enumerate(cams):
        ei, n = cross_camera_graph(c)
        x, ea = inputs(n, ei.shape[1], 100 + k)
        out.append((x, ei, ea))
    return out


def _gpu(f):
    d = Data()
    d.x, d.edge_index, d.edge_attr = (torch.from_numpy(t).cuda() for t in f)
    return d


def test_pad_kernel_writes_the_canonical_frame():
    from gnn_cca_amd import _native as nat
    x, ei, ea = _frames()[1]
    n, e = x.shape[0], ei.shape[1]
    n_max, n_dummy, e_pad = 12, 3, e + 10
    d = _gpu((x, ei, ea))
    xp = torch.full((n_max + n_dummy, x.shape[1]), 7.0, device="cuda")
    eip = torch.full((2, e_pad), -1, dtype=torch.int64, device="cuda")
    eap = torch.full((e_pad, ea.shape[1]), 7.0, device="cuda")
    nat.check(nat.lib().gnncca_pad_frame(d.x.data_ptr(), n, d.edge_index.data_ptr(), d.edge_attr.data_ptr(), e, xp.data_ptr(), n_max, n_dummy,
                                         eip.data_ptr(), eap.data_ptr(), e_pad, x.shape[1], ea.shape[1],
                                         torch.cuda.current_stream().cuda_stream), "pad")
    torch.cuda.synchronize()
    assert torch.equal(xp[:n].cpu(), torch.from_numpy(x)) and float(xp[n:].abs().max()) == 0.0
    assert torch.equal(eap[:e].cpu(), torch.from_numpy(ea)) and float(eap[e:].abs().max()) == 0.0
    got = eip.cpu().numpy()
    assert np.array_equal(got[:, :e], ei)
    pad = got[:, e:]
    assert np.array_equal(pad[0], pad[1])                                    # self loops
    assert pad.min() >= n_max and pad.max() < n_max + n_dummy                # on the dummy nodes only
    assert np.all(np.diff(got[0]) >= 0)                                      # rows stay sorted
    assert np.bincount(pad[0] - n_max, minlength=n_dummy).max() == 4         # dealt over the dummy nodes: ceil(10 / 3)
    # refusals: a frame that does not fit, no dummy node
    bad = nat.lib().gnncca_pad_frame(d.x.data_ptr(), n, d.edge_index.data_ptr(), d.edge_attr.data_ptr(), e, xp.data_ptr(), n - 1, n_dummy,
                                     eip.data_ptr(), eap.data_ptr(), e_pad, x.shape[1], ea.shape[1], 0)
    assert bad == nat.ERR_INVALID_ARG
    bad = nat.lib().gnncca_pad_frame(d.x.data_ptr(), n, d.edge_index.data_ptr(), d.edge_attr.data_ptr(), e, xp.data_ptr(), n_max, 0,
                                     eip.data_ptr(), eap.data_ptr(), e_pad, x.shape[1], ea.shape[1], 0)
    assert bad == nat.ERR_INVALID_ARG


@pytest.mark.parametrize("streams", [1, 3])
def test_one_graph_serves_every_frame_shape(streams):
    from gnn_cca_amd.inference import PaddedForward
    m, params, arch, sd, _ = _model()
    frames = _frames()
    n_max = max(f[0].shape[0] for f in frames)
    e_max = max(f[1].shape[1] for f in frames)
    oracle = NumpyOracle(params, arch, sd, np.float32)
    pf = PaddedForward(m, n_max=n_max, e_max=e_max, n_dummy=4, streams=streams)
    for rep in range(2):
        for lo in range(0, len(frames), streams):
            chunk = frames[lo:lo + streams]
            res = [pf(_gpu(f)) for f in chunk]                              # `streams` frames in flight
            for f, r in zip(chunk, res):
                out = r.result() if streams > 1 else r
                got = [t.clone() for t in out["classified_edges"]]
                with torch.no_grad():
                    want = m(_gpu(f))["classified_edges"]
                ref = oracle.forward(*f)
                assert len(got) == len(want) == len(ref)
                for g, w, r_ in zip(got, want, ref):
                    assert g.shape == w.shape == (f[1].shape[1], 1)
                    assert float((g - w).abs().max()) <= 5e-6
                    assert np.abs(g.cpu().numpy() - r_).max() <= TOL
    assert pf.padded == 2 * len(frames) and pf.eager == 0
    assert len(pf._gf._graphs) == streams                                    # one graph per stream, whatever the frame shape


def test_frames_that_do_not_fit_run_eagerly_and_weight_updates_are_seen():
    from gnn_cca_amd.inference import PaddedForward
    m, params, arch, sd, _ = _model()
    frames = _frames()
    small, big = frames[1], frames[6]
    pf = PaddedForward(m, n_max=small[0].shape[0], e_max=small[1].shape[1])   # exactly the small frame: no padding edge at all
    with torch.no_grad():
        want_s = [t.clone() for t in m(_gpu(small))["classified_edges"]]
        want_b = [t.clone() for t in m(_gpu(big))["classified_edges"]]
    got = pf(_gpu(small))["classified_edges"]
    for g, w in zip(got, want_s):
        assert float((g - w).abs().max()) <= 5e-6
    got = pf(_gpu(big))["classified_edges"]
    for g, w in zip(got, want_b):
        assert torch.equal(g, w)
    assert (pf.padded, pf.eager) == (1, 1)
    empty = Data()
    empty.x, empty.edge_index, empty.edge_attr = _gpu(small).x[:3], torch.zeros((2, 0), dtype=torch.int64, device="cuda"), torch.zeros((0, 4), device="cuda")
    assert pf(empty)["classified_edges"][-1].shape == (0, 1)
    # a parameter update between frames is seen by the next replay (the packed weights are rebuilt in place)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(0.9)
        want2 = [t.clone() for t in m(_gpu(small))["classified_edges"]]
    got = pf(_gpu(small))["classified_edges"]
    for g, w, w_old in zip(got, want2, want_s):
        assert float((g - w).abs().max()) <= 5e-6
    assert float((want2[-1] - want_s[-1]).abs().max()) > 1e-4
    m.train()
    with pytest.raises(RuntimeError):
        pf(_gpu(small))
    m.eval()
    # a frame of another feature width is refused, not padded from the wrong rows
    bad = _gpu(small)
    bad.x = bad.x[:, :100].contiguous()
    with pytest.raises(RuntimeError):
        pf(bad)
    bad = _gpu(small)
    bad.edge_attr = bad.edge_attr[:-1]
    with pytest.raises(RuntimeError):
        pf(bad)

"""gnn_cca_amd.inference.GraphedForward: HIP-graph replays of the eval forward (per frame, K frames per graph, S streams) give
BITWISE the eager logits -- the per-frame caller of inference.py:173-283 on the product path bench.py times."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle.mpn_oracle import NumpyOracle, load_case  # the checker

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Data:
    pass


def _model(case="terrace32"):
    from gnn_cca_amd import MOTMPNet
    params, arch, sd, a = load_case(os.path.join(GOLDEN, case + ".npz"))
    m = MOTMPNet(copy.deepcopy(params), None, arch)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.cuda().eval(), params, arch, sd, a


def _frame(a, seed=None):
    d = Data()
    d.x, d.edge_index, d.edge_attr = (torch.from_numpy(np.asarray(a[k])).cuda() for k in ("x", "edge_index", "edge_attr"))
    if seed is not None:   # another frame of the same shape
        g = torch.Generator().manual_seed(seed)
        d.x = torch.nn.functional.normalize(torch.randn(tuple(d.x.shape), generator=g), p=2, dim=0).cuda()
        d.edge_attr = torch.rand(tuple(d.edge_attr.shape), generator=g).cuda()
    return d


def _dense(n, seed):
    import bench
    return bench.make_data(n, 1, seed, "cuda")


def _eq(got, want):
    return len(got) == len(want) and all(torch.equal(g, w) for g, w in zip(got, want))


def test_per_frame_replay_is_bitwise_eager_and_matches_the_golden():
    from gnn_cca_amd.inference import GraphedForward
    m, params, arch, sd, a = _model()
    gf = GraphedForward(m, warmup=2)
    frames = [_frame(a)] + [_frame(a, seed=s) for s in range(1, 7)]
    with torch.no_grad():
        want = [[t.clone() for t in m(f)["classified_edges"]] for f in frames]
    for i, f in enumerate(frames):       # 2 eager calls, 1 capture, 4 replays -- a different frame every time
        got = gf(f)["classified_edges"]
        assert _eq(got, want[i]), i
    assert len(gf._graphs) == 1
    # frame 0 is the golden case: the replayed logits are the reference's within the usual bound
    got0 = gf(frames[0])["classified_edges"]
    for i, t in enumerate(got0):
        assert np.abs(t.cpu().numpy() - a[f"logits_{i}"]).max() <= 5e-6
    ref = NumpyOracle(params, arch, sd, np.float32).forward(a["x"], a["edge_index"], a["edge_attr"])
    assert max(float(np.abs(t.cpu().numpy() - r).max()) for t, r in zip(got0, ref)) <= 1e-5


def test_frames_written_into_the_static_inputs_replay_without_copies():
    from gnn_cca_amd.inference import GraphedForward
    m, *_, a = _model()
    gf = GraphedForward(m, warmup=0)
    f0 = _frame(a)
    st = gf.static_inputs(f0)
    assert st is gf.static_inputs(f0)
    st.x.copy_(f0.x), st.edge_index.copy_(f0.edge_index), st.edge_attr.copy_(f0.edge_attr)
    out = gf(st)["classified_edges"]                      # captured on the handed-out buffers
    assert gf.static_inputs(f0) is st
    with torch.no_grad():
        assert _eq(out, m(f0)["classified_edges"])
    f1 = _frame(a, seed=5)
    st.x.copy_(f1.x), st.edge_attr.copy_(f1.edge_attr)    # the producer writes the next frame in place
    out = gf(st)["classified_edges"]
    with torch.no_grad():
        assert _eq(out, m(f1)["classified_edges"])


def test_shapes_get_their_own_graphs_and_weight_updates_are_seen():
    from gnn_cca_amd.inference import GraphedForward
    m, *_ = _model()
    gf = GraphedForward(m, warmup=1)
    shapes = [_dense(n, 10 + n) for n in (12, 40, 64)]
    with torch.no_grad():
        for rep in range(4):
            for d in shapes:
                assert _eq(gf(d)["classified_edges"], m(d)["classified_edges"]), (rep, d.x.shape)
        assert len(gf._graphs) == 3
        # in-place parameter update (an optimizer step, load_state_dict): the next call repacks and the replay uses the new weights
        for p in m.classifier.parameters():
            p.mul_(1.5)
        want = [t.clone() for t in m(shapes[1])["classified_edges"]]
        assert _eq(gf(shapes[1])["classified_edges"], want)
        sd = {k: v * 0.5 if v.dtype.is_floating_point else v for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        want = [t.clone() for t in m(shapes[2])["classified_edges"]]
        assert _eq(gf(shapes[2])["classified_edges"], want)
        # an option that selects other kernels invalidates the captured graphs instead of replaying the old ones
        m.edge_state_dtype = "bf16"
        want = [t.clone() for t in m(shapes[2])["classified_edges"]]
        assert _eq(gf(shapes[2])["classified_edges"], want)


def test_block_of_k_frames_in_one_graph():
    from gnn_cca_amd.inference import GraphedForward
    m, *_, a = _model()
    gf = GraphedForward(m)
    frames = [_frame(a, seed=s) for s in range(4)] + [_dense(48, 3)]   # mixed shapes in one block
    with torch.no_grad():
        want = [[t.clone() for t in m(f)["classified_edges"]] for f in frames]
    blk = gf.block(frames)
    assert gf.block(frames) is blk and len(blk) == 5
    outs = blk.replay()
    for o, w in zip(outs, want):
        assert _eq(o["classified_edges"], w)
    # new frames of the same shapes through the static inputs
    frames2 = [_frame(a, seed=20 + s) for s in range(4)] + [_dense(48, 9)]
    with torch.no_grad():
        want2 = [[t.clone() for t in m(f)["classified_edges"]] for f in frames2]
    outs = blk.replay(frames2)
    for o, w in zip(outs, want2):
        assert _eq(o["classified_edges"], w)
    with pytest.raises(ValueError):
        blk.replay(frames2[:3])
    # a block notices that it was captured for other kernels (an option that changes the dispatch) instead of replaying them
    m.edge_state_dtype = "bf16"
    with pytest.raises(RuntimeError):
        blk.replay()
    m.edge_state_dtype = "fp32"
    outs = blk.replay(frames2)
    for o, w in zip(outs, want2):
        assert _eq(o["classified_edges"], w)
    # the bench form: ONE resident frame K times, inputs adopted (no copies), every forward with its own outputs
    d = _dense(64, 1)
    blk = gf.block([d] * 6, adopt_inputs=True)
    assert blk.inputs[0].x.data_ptr() == d.x.data_ptr()
    outs = blk.replay()
    with torch.no_grad():
        want = m(d)["classified_edges"]
    ptrs = {o["classified_edges"][-1].data_ptr() for o in outs}
    assert len(ptrs) == 6
    for o in outs:
        assert _eq(o["classified_edges"], want)
    # the same K forwards on three parallel branches of one graph (three frames in flight, a workspace per branch)
    frames3 = [_frame(a, seed=40 + s) for s in range(7)]
    with torch.no_grad():
        want3 = [[t.clone() for t in m(f)["classified_edges"]] for f in frames3]
    blk3 = gf.block(frames3, chains=3, depth=2)      # graphs of two frames, round robin on three streams
    assert len(blk3._graphs) == 6        # a multiple of the stream count: 2 + 1 + 1 + 1 + 1 + 1 frames, i.e. 3 / 2 / 2 frames per stream
    for rep in range(3):
        outs = blk3.replay()
        for o, w in zip(outs, want3):
            assert _eq(o["classified_edges"], w), rep


def test_forwards_in_flight_on_several_streams():
    from gnn_cca_amd.inference import GraphedForward
    m, *_, a = _model()
    gf = GraphedForward(m, streams=3)
    frames = [_frame(a, seed=s) for s in range(9)] + [_dense(40, s) for s in range(3)]
    with torch.no_grad():
        want = [[t.clone() for t in m(f)["classified_edges"]] for f in frames]
    torch.cuda.synchronize()
    for lo in range(0, len(frames), 3):      # three in flight, then collect: a slot's outputs are reused by its next frame
        pend = [gf.submit(f) for f in frames[lo:lo + 3]]
        for i, p in enumerate(pend):
            got = [t.clone() for t in p.result()["classified_edges"]]
            assert _eq(got, want[lo + i]), lo + i
    # frames written into a slot's own static inputs replay on that slot without copies
    slots = [gf.slot_inputs(frames[0], i) for i in range(3)]
    assert len({s.x.data_ptr() for s in slots}) == 3
    for i, s in enumerate(slots):
        s.x.copy_(frames[i].x), s.edge_attr.copy_(frames[i].edge_attr)
    torch.cuda.synchronize()
    pend = [gf.submit(s, after_current=False) for s in slots]
    for i, p in enumerate(pend):
        assert _eq(p.result()["classified_edges"], want[i]), i
    # the lean form: the producer refills the slots and replays them; join() orders the current stream behind all of them
    for i, s in enumerate(slots):
        s.x.copy_(frames[3 + i].x), s.edge_attr.copy_(frames[3 + i].edge_attr)
    torch.cuda.synchronize()
    for i in range(3):
        gf.replay_slot(i)
    gf.join()
    for i in range(3):
        assert _eq(gf.slot_outputs(i)["classified_edges"], want[3 + i]), i
    with pytest.raises(RuntimeError):
        GraphedForward(m).submit(frames[0])
    # a change of the module's options (or of the blob's address) drops every graph captured under the old ones, and a slot
    # whose graph went with them must refuse to replay it (it addressed the dropped workspace and blob) instead of reading freed memory
    m.edge_state_dtype = "bf16"
    try:
        gf.submit(frames[0]).result()
        torch.cuda.synchronize()
        refused = 0
        for i in range(3):
            try:
                gf.replay_slot(i)
            except RuntimeError:
                refused += 1
        gf.join()
        torch.cuda.synchronize()
        assert refused == 2      # the slot that took the new frame replays its NEW graph; the other two have nothing live
    finally:
        m.edge_state_dtype = "fp32"


def test_refusals():
    from gnn_cca_amd.inference import GraphedForward
    m, *_, a = _model()
    gf = GraphedForward(m)
    d = _frame(a)
    cpu = Data()
    cpu.x, cpu.edge_index, cpu.edge_attr = d.x.cpu(), d.edge_index.cpu(), d.edge_attr.cpu()
    with pytest.raises(RuntimeError):
        gf(cpu)
    bad = Data()
    bad.x, bad.edge_index, bad.edge_attr = d.x.double(), d.edge_index, d.edge_attr
    with pytest.raises(RuntimeError):
        gf(bad)
    m.train()
    with pytest.raises(RuntimeError):
        gf(d)
    m.eval()
    empty = Data()
    empty.x, empty.edge_index, empty.edge_attr = d.x, d.edge_index[:, :0].contiguous(), d.edge_attr[:0].contiguous()
    out = gf(empty)["classified_edges"]
    assert all(t.shape == (0, 1) for t in out)


def test_sharded_forward_as_one_graph_replay_per_step():
    """sharding.forward_sharded(..., graphed=GraphedForward): a rank's resident union replayed as ONE HIP graph captured on the union's own
    tensors -- bit for bit the eager forward_sharded, replay after replay, and a union overwritten IN PLACE is seen by the next replay
    (no copies, no stale inputs).  Two shares of one sequence through one GraphedForward (what bench.py's config-4 legs do)."""
    from gnn_cca_amd.inference import GraphedForward
    from gnn_cca_amd.sharding import forward_sharded, shard_batch
    m, *_ = _model("dense64")
    graphs = []
    for g, n in enumerate((24, 64, 33, 64, 12, 40, 64, 50)):
        d = _dense(n, 100 + g)
        graphs.append((d.x, d.edge_index, d.edge_attr))
    gf = GraphedForward(m, warmup=0)
    with torch.no_grad():
        for rank, world in ((0, 1), (1, 3)):
            lo, hi, batch = shard_batch(graphs, rank, world)
            want = [[t.clone() for t in steps] for steps in forward_sharded(m, graphs, rank, world, batch=batch)[2]]
            for _ in range(3):
                lo2, hi2, got = forward_sharded(m, graphs, rank, world, batch=batch, graphed=gf)
                assert (lo2, hi2) == (lo, hi) and len(got) == hi - lo
                for a, b in zip(got, want):
                    assert _eq(a, b)
            batch.edge_attr.mul_(0.5)              # the producer overwrites the resident union in place
            want2 = [[t.clone() for t in steps] for steps in forward_sharded(m, graphs, rank, world, batch=batch)[2]]
            _, _, got2 = forward_sharded(m, graphs, rank, world, batch=batch, graphed=gf)
            for a, b, c in zip(got2, want2, want):
                assert _eq(a, b)
                assert not torch.equal(b[-1], c[-1])

"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding logic and the weight-blob broadcast.
(No compute: the forward itself is HIP-only.)"""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR
from oracle.mpn_oracle import load_case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_cca_amd import MOTMPNet
        from gnn_cca_amd.sharding import broadcast_packed_weights, shard_range
        params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "terrace32.npz"))
        torch.manual_seed(100 + rank)  # ranks start from DIFFERENT random weights
        m = MOTMPNet(copy.deepcopy(params), None, arch).eval()
        if rank == 0:
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        blob = broadcast_packed_weights(m, src=0)
        covered = list(range(*shard_range(7, rank, world)))
        # the broadcast lands in this rank's OWN tensors: dropping the packed cache (.train() / .eval()) and packing again
        # must give rank 0's blob, not this rank's initial weights
        m.train()
        m.eval()
        again = m.pack_weights_host().numpy().tobytes()
        sd_bytes = b"".join(v.numpy().tobytes() for v in m.state_dict().values())
        q.put((rank, blob.numpy().tobytes(), covered, again, sd_bytes))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "every rank must hold rank 0's packed weights after the broadcast"
    assert res[0][2] + res[1][2] == list(range(7))
    assert res[1][3] == res[0][1], "repacking after .train()/.eval() on rank 1 must still give rank 0's weights"
    assert res[1][4] == res[0][4], "state_dict of rank 1 must be rank 0's after the broadcast"
    # and the blob is rank 0's: equal to packing the golden state_dict locally
    from gnn_cca_amd import MOTMPNet
    params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "terrace32.npz"))
    m = MOTMPNet(copy.deepcopy(params), None, arch).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    assert m.pack_weights_host().numpy().tobytes() == res[0][1]


@pytest.mark.parametrize("n,world", [(512, 8), (7, 3), (2, 4), (0, 2), (64, 1)])
def test_shard_range_partitions(n, world):
    from gnn_cca_amd.sharding import shard_range
    got = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
        got += list(range(lo, hi))
    assert got == list(range(n))


def test_union_graphs_layout():
    from gnn_cca_amd.sharding import split_logits, union_graphs
    g1 = (torch.zeros(3, 5), torch.tensor([[0, 0, 1], [1, 2, 2]]), torch.zeros(3, 4))
    g2 = (torch.ones(2, 5), torch.tensor([[0, 1], [1, 0]]), torch.ones(2, 4))
    b = union_graphs([g1, g2])
    assert b.x.shape == (5, 5) and b.edge_attr.shape == (5, 4)
    assert b.edge_index.tolist() == [[0, 0, 1, 3, 4], [1, 2, 2, 4, 3]]
    assert b.edge_ptr == [0, 3, 5] and b.node_ptr == [0, 3, 5]
    parts = split_logits({'classified_edges': [torch.arange(5.).view(5, 1)]}, b)
    assert [p[0].view(-1).tolist() for p in parts] == [[0., 1., 2.], [3., 4.]]


def test_forward_sharded_host_logic():
    """forward_sharded / shard_batch with a stand-in model (the real forward is HIP-only): every graph is covered once
    over the ranks, a prebuilt union is reused, a lazy sequence is only sliced by the rank's range."""
    from gnn_cca_amd.sharding import forward_sharded, shard_batch

    class Echo:  # 'logit' of an edge = its global source node id inside the rank's union
        def __call__(self, batch):
            return {'classified_edges': [batch.edge_index[0].float().view(-1, 1)]}

    class Lazy:
        def __init__(self, graphs):
            self.graphs, self.asked = graphs, []

        def __len__(self):
            return len(self.graphs)

        def __getitem__(self, sl):
            self.asked.append((sl.start, sl.stop))
            return self.graphs[sl]

    graphs = [(torch.zeros(n, 3), torch.tensor([[i for i in range(n)], [(i + 1) % n for i in range(n)]]), torch.zeros(n, 4))
              for n in (2, 3, 4, 5, 6)]
    seen = []
    for rank in range(3):
        lazy = Lazy(graphs)
        lo, hi, batch = shard_batch(lazy, rank, 3)
        assert lazy.asked == [(lo, hi)]
        lo2, hi2, per_graph = forward_sharded(Echo(), lazy, rank, 3, batch=batch)
        assert (lo, hi) == (lo2, hi2) and lazy.asked == [(lo, hi)]  # the union was not rebuilt
        assert len(per_graph) == hi - lo
        for g, steps in zip(range(lo, hi), per_graph):
            n = graphs[g][0].shape[0]
            assert steps[0].shape == (n, 1)
            seen.append(g)
    assert seen == list(range(5))
    assert forward_sharded(Echo(), graphs[:1], 1, 2) == (1, 1, [])
    with pytest.raises(ValueError):
        forward_sharded(Echo(), graphs, 0, 3, batch=shard_batch(graphs[:1], 0, 1)[2])  # rank 0 of 3 owns two graphs

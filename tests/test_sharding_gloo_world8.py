"""The N = 8 operating point of SURVEY.md 8(e) rehearsed on CPU: EIGHT gloo rank processes run everything of the multi-GPU path that is
host logic -- `shard_range` / `shard_batch` over BASELINE config 4's 512 graphs (tiny ones here), the ONE weight broadcast
(`broadcast_weights`: every rank starts from different random weights and must end with rank 0's state_dict and packed blob),
`forward_sharded` with a stand-in model (the real forward is HIP-only) -- and `bench.py`'s agreement step: a HIP-graph capture that fails on
ONE rank must drop that form on EVERY rank, otherwise the timed blocks' barriers pair up across different forms.
The world-2 twin is tests/test_sharding_gloo.py."""
import copy
import hashlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR
from oracle.mpn_oracle import load_case

WORLD, GRAPHS = 8, 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _LazyTinyGraphs:
    """512 ring graphs of 2-5 nodes, graph g seeded by g alone (the same on every rank), materialised only for the slice asked for."""

    def __init__(self, n):
        self.n, self.asked = n, []

    def __len__(self):
        return self.n

    def __getitem__(self, sl):
        self.asked.append((sl.start, sl.stop))
        out = []
        for g in range(*sl.indices(self.n)):
            k = 2 + g % 4
            gen = torch.Generator().manual_seed(9000 + g)
            ei = torch.tensor([list(range(k)), [(i + 1) % k for i in range(k)]])
            out.append((torch.randn(k, 8, generator=gen), ei, torch.rand(k, 4, generator=gen)))
        return out


class _Echo:
    """Stand-in forward: an edge's 'logit' = 1000 * (its graph's GLOBAL id) + its position inside the graph -- so the parent can check that
    every graph came back from the rank that owns it, once, in order."""

    def __init__(self, lo):
        self.lo = lo

    def __call__(self, batch):
        vals = []
        for i in range(len(batch.edge_ptr) - 1):
            k = batch.edge_ptr[i + 1] - batch.edge_ptr[i]
            vals.append(1000.0 * (self.lo + i) + torch.arange(k, dtype=torch.float32))
        return {"classified_edges": [torch.cat(vals).view(-1, 1)]}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from gnn_cca_amd import MOTMPNet
        from gnn_cca_amd.sharding import broadcast_weights, forward_sharded, shard_batch, shard_range
        params, arch, sd, _ = load_case(os.path.join(GOLDEN_DIR, "terrace32.npz"))
        torch.manual_seed(500 + rank)                      # ranks start from DIFFERENT random weights
        m = MOTMPNet(copy.deepcopy(params), None, arch).eval()
        if rank == 0:
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        before = hashlib.blake2b(b"".join(v.numpy().tobytes() for v in m.state_dict().values()), digest_size=8).hexdigest()
        broadcast_weights(m, src=0)
        after = hashlib.blake2b(b"".join(v.numpy().tobytes() for v in m.state_dict().values()), digest_size=8).hexdigest()
        blob = hashlib.blake2b(m.pack_weights_host().numpy().tobytes(), digest_size=8).hexdigest()
        # bench.py's own verification collective (hash all-gather) on this backend
        mine = bench.state_hash(m)
        hashes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(hashes, torch.tensor([mine], dtype=torch.int64))
        # sharding: config 4's 512 graphs over 8 ranks, lazily built, the resident union reused
        graphs = _LazyTinyGraphs(GRAPHS)
        lo, hi, batch = shard_batch(graphs, rank, world)
        assert (lo, hi) == shard_range(GRAPHS, rank, world) and graphs.asked == [(lo, hi)]
        lo2, hi2, per_graph = forward_sharded(_Echo(lo), graphs, rank, world, batch=batch)
        assert (lo2, hi2) == (lo, hi) and graphs.asked == [(lo, hi)] and len(per_graph) == hi - lo
        firsts = [int(steps[0][0, 0].item()) // 1000 for steps in per_graph]
        sizes = [int(steps[0].shape[0]) for steps in per_graph]
        # the agreement step of bench.py: rank 3 "failed" to capture the graph forms, rank 5 only the chains
        have = {"eager": 1, "graph": 1, "graph_block": 1, "graph_block_chains": 1}
        if rank == 3:
            have = {"eager": 1}
        if rank == 5:
            have.pop("graph_block_chains")
        agreed = bench.agree_forms(have, dist, torch.device("cpu"), "gloo")
        all_have = bench.agree_forms({"eager": 1, "graph": 1}, dist, torch.device("cpu"), "gloo")
        flag_all = bench.agree_flag(True, dist, torch.device("cpu"), "gloo")
        flag_one_off = bench.agree_flag(rank != 6, dist, torch.device("cpu"), "gloo")
        q.put((rank, before, after, blob, [int(h.item()) for h in hashes], (lo, hi), firsts, sizes, agreed, all_have, flag_all, flag_one_off))
    finally:
        dist.destroy_process_group()


def test_world8_broadcast_sharding_and_form_agreement():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    befores = {r[1] for r in res}
    assert len(befores) == WORLD, "ranks must START from different weights (otherwise the broadcast proves nothing)"
    assert {r[2] for r in res} == {res[0][2]} and res[0][2] == res[0][1], "every rank holds rank 0's state_dict after the ONE broadcast"
    assert {r[3] for r in res} == {res[0][3]}, "every rank packs the same kernel blob from it"
    assert all(len(set(r[4])) == 1 for r in res), "bench.py's hash all-gather sees one hash on every rank"
    covered = []
    for rank, *_rest in res:
        lo, hi = _rest[4]
        assert hi - lo == GRAPHS // WORLD                   # 64 graphs per rank: BASELINE config 4's split
        assert _rest[5] == list(range(lo, hi))              # every graph from the rank that owns it, in order
        assert _rest[6] == [2 + g % 4 for g in range(lo, hi)]
        covered += list(range(lo, hi))
    assert covered == list(range(GRAPHS))
    for r in res:
        assert r[8] == ["eager"], "a form one rank could not capture is dropped on EVERY rank"
        assert r[9] == ["eager", "graph"] and r[10] is True and r[11] is False

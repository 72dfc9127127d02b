"""One rank of the two-rank sharded-forward test (tests/test_gpu_sharded.py starts two FRESH copies of this script).

    python sharded_worker.py <rank> <world> <port> <outdir>

Backend: RCCL ("nccl") when the box shows >= `world` GPUs (one per rank), otherwise gloo with every rank on cuda:0.
Rank 0 loads the golden weights; every other rank starts from different random weights and must end up with rank 0's
after gnn_cca_amd.sharding.broadcast_weights.  Each rank runs forward_sharded over the same 7 unequal graphs and writes
the logits of the graphs it owns.
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SIZES = (5, 9, 12, 17, 23, 30, 40)


def make_graphs(device):
    """7 dense directed graphs of unequal size with local node ids; identical on every rank (seeded per graph)."""
    graphs = []
    for g, n in enumerate(SIZES):
        gen = torch.Generator().manual_seed(4242 + g)
        x = torch.nn.functional.normalize(torch.randn(n, 2048, generator=gen), p=2, dim=0)
        i = torch.arange(n).repeat_interleave(n)
        j = torch.arange(n).repeat(n)
        keep = i != j
        ei = torch.stack([i[keep], j[keep]])
        ea = torch.rand(ei.shape[1], 4, generator=gen)
        graphs.append((x.to(device), ei.to(device), ea.to(device)))
    return graphs


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    multi = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    backend = "nccl" if multi else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_cca_amd import MOTMPNet
        from gnn_cca_amd.sharding import broadcast_weights, forward_sharded
        from oracle.mpn_oracle import load_case  # fixture loader only (params + golden weights)
        params, arch, sd, _ = load_case(os.path.join(ROOT, "tests", "golden", "terrace32.npz"))
        torch.manual_seed(100 + rank)
        m = MOTMPNet(copy.deepcopy(params), None, arch)
        if rank == 0:
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.to(dev)
        broadcast_weights(m, src=0)
        m.eval()  # drops the packed cache: the next forward must repack RANK 0's weights, not this rank's initial ones
        graphs = make_graphs(dev)
        with torch.no_grad():
            lo, hi, per_graph = forward_sharded(m, graphs, rank, world)
        torch.cuda.synchronize()
        out = {"lo": lo, "hi": hi, "backend": backend}
        for g, steps in zip(range(lo, hi), per_graph):
            for s, t in enumerate(steps):
                out[f"g{g}_s{s}"] = t.cpu().numpy()
        for k, v in m.state_dict().items():
            out["sd::" + k] = v.cpu().numpy()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

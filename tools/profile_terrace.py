"""GPU box: where the host time of bench.py's `terrace_pipeline` loop goes (cProfile over the pipelined loop), and the loop's rate.
    python tools/profile_terrace.py [n_batches=16] [reps=5]"""
import copy
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from gnn_cca_amd.graph_build import build_graph_batch
    from gnn_cca_amd.postprocess import prune_and_cluster, threshold
    dev = torch.device("cuda:0")
    frames = bench.terrace_frames(64, n_batches)
    model = bench.build_model(copy.deepcopy(bench.graph_net_params(L=4)), 20, seed=0).to(dev).eval()
    dev_in = [(torch.from_numpy(f["node"]).to(dev), torch.from_numpy(f["reid"]).to(dev)) for f in frames]

    def run(i):
        f, (node, reid) = frames[i], dev_in[i]
        b = build_graph_batch(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
        with torch.no_grad():
            out = model(b)
        probs, preds = threshold(out["classified_edges"][-1])
        return prune_and_cluster(b.edge_index, preds, b.x.shape[0], b.node_ptr_dev, b.edge_ptr_dev)

    from gnn_cca_amd.pipeline import FramePipeline
    pipe = FramePipeline(model)

    def run_pipe(i):
        f, (node, reid) = frames[i], dev_in[i]
        return pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)

    if os.environ.get("AB"):   # both forms in ONE process, alternating (separate short processes see different GPU clock states)
        for fn in (run, run_pipe):
            for i in range(n_batches):
                fn(i)
        torch.cuda.synchronize()
        for rnd in range(4):
            for name, fn in (("step by step ", run), ("FramePipeline", run_pipe)):
                t0 = time.perf_counter()
                for _ in range(reps):
                    for i in range(n_batches):
                        fn(i)
                torch.cuda.synchronize()
                print(f"{name}: {(time.perf_counter() - t0) / (reps * n_batches) * 1e3:.4f} ms per batch", flush=True)
        return
    if os.environ.get("PIPE"):
        run = run_pipe
    for i in range(n_batches):
        run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(n_batches):
            run(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"ms per batch {dt / (reps * n_batches) * 1e3:.4f} (host loop alone {t_host / (reps * n_batches) * 1e3:.4f})")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n_batches):
        run(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:34]
    print("us per batch: tottime  cumtime  calls  function")
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print(f"  {tt / n_batches * 1e6:8.1f} {ct / n_batches * 1e6:8.1f} {nc / n_batches:6.1f}  {os.path.basename(fn)}:{line}({name})")


if __name__ == "__main__":
    main()

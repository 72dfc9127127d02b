"""GPU box: the reference's per-frame inference loop (inference.py:173-283: one forward per frame, batch size 1) over real-shaped
Terrace frames (tests/golden/terrace_topology.npz, synthetic features), MPN forward only, frames resident in HBM:
  eager            model(data) per frame
  graph_per_shape  GraphedForward: a HIP graph per frame shape (hundreds of shapes in a sequence)
  padded           PaddedForward: ONE graph, every frame padded to (34 + 8 nodes, 866 edges) by gnncca_pad_frame
  padded_x3        the same with three frames in flight
    python tools/per_frame_terrace.py [n_frames=1024]
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    from gnn_cca_amd.graph_build import build_graph_batch
    from gnn_cca_amd.inference import GraphedForward, PaddedForward
    dev = torch.device("cuda:0")
    model = bench.build_model(copy.deepcopy(bench.graph_net_params(L=4)), 20, seed=0).to(dev).eval()
    frames = []
    for f in bench.terrace_frames(1, n_frames):
        node, reid = torch.from_numpy(f["node"]).to(dev), torch.from_numpy(f["reid"]).to(dev)
        b = build_graph_batch(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
        if b.edge_index.shape[1] > 0:
            frames.append(b)
    torch.cuda.synchronize()
    shapes = {(int(b.x.shape[0]), int(b.edge_index.shape[1])) for b in frames}
    n_max, e_max = max(s[0] for s in shapes), max(s[1] for s in shapes)
    with torch.no_grad():
        want = [model(b)["classified_edges"][-1].clone() for b in frames]

    def timed(fn, reps=3):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best / len(frames) * 1e6

    res = {"frames": len(frames), "distinct_shapes": len(shapes), "n_max": n_max, "e_max": e_max,
           "edges_per_frame": float(np.mean([b.edge_index.shape[1] for b in frames]))}

    def eager():
        with torch.no_grad():
            for b in frames:
                model(b)
    res["eager_us_per_frame"] = timed(eager)

    gf = GraphedForward(model, warmup=0, max_graphs=4096)
    t0 = time.perf_counter()
    for b in frames:
        gf(b)
    torch.cuda.synchronize()
    res["graph_per_shape_capture_s"] = time.perf_counter() - t0
    res["graph_per_shape_graphs"] = len(gf._graphs)
    res["graph_per_shape_us_per_frame"] = timed(lambda: [gf(b) for b in frames])
    del gf

    for s in (1, 3):
        pf = PaddedForward(model, n_max=n_max, e_max=e_max, n_dummy=8, streams=s)
        worst = 0.0
        for lo in range(0, len(frames), s):
            outs = [pf(b) for b in frames[lo:lo + s]]
            for i, o in enumerate(outs):
                o = o.result() if s > 1 else o
                worst = max(worst, float((o["classified_edges"][-1] - want[lo + i]).abs().max()))
        torch.cuda.synchronize()

        def run():
            for b in frames:
                pf(b)
            if s > 1:
                pf.join()
        res["padded%s_us_per_frame" % ("" if s == 1 else "_x%d" % s)] = timed(run)
        if os.environ.get("PROFILE"):
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            run()
            torch.cuda.synchronize()
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(14)
        res["padded%s_max_abs_vs_eager" % ("" if s == 1 else "_x%d" % s)] = worst
        assert pf.eager == 0
    print(json.dumps(res))


if __name__ == "__main__":
    main()

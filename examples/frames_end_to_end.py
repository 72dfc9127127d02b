#!/usr/bin/env python3
"""End-to-end walk through the reference's per-batch inference flow (inference.py:173-345) on synthetic frames, with
every GPU-side step taken by this repository: graph construction + edge attributes (row N1), the MPN forward (the hot
path), threshold / pruning / identity clusters (row N2).  Needs an MI355X.

    python examples/frames_end_to_end.py [frames] [cams] [detections_per_cam]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (model / GRAPH_NET_PARAMS builders)
from gnn_cca_amd.graph_build import build_graph_batch  # noqa: E402
from gnn_cca_amd.postprocess import prune_and_cluster, threshold  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(0)
    n_g = cams * per
    n = frames * n_g
    # what libs/datasets.py would hand over per frame: detections with camera id, person id, ground-plane position
    id_cam = np.tile(np.repeat(np.arange(cams), per), frames)
    ids = np.concatenate([rng.integers(0, per, size=n_g) for _ in range(frames)]).astype(np.int64)
    pos = rng.uniform(-8, 8, size=(frames, per, 2))
    frame_of = np.repeat(np.arange(frames), n_g)
    xw = pos[frame_of, ids, 0] + rng.normal(0, 0.3, n)
    yw = pos[frame_of, ids, 1] + rng.normal(0, 0.3, n)
    max_dist = [80.0] * frames                       # CONFIG['CONV_TO_M'][dataset]
    node_embeds = torch.randn(n, 2048, device="cuda")  # ReID CNN outputs (inference.py:183), random here
    reid_embeds = torch.randn(n, 256, device="cuda")
    model = bench.build_model(bench.graph_net_params(), n_g).cuda().eval()

    def run():
        batch = build_graph_batch(xw, yw, ids, id_cam, [n_g] * frames, max_dist, node_embeds, reid_embeds)
        with torch.no_grad():
            out = model(batch)
        probs, preds = threshold(out["classified_edges"][-1])
        post = prune_and_cluster(batch.edge_index, preds, n, batch.node_ptr_dev, batch.edge_ptr_dev)
        return batch, probs, post

    # random weights put every logit on one side of 0; centre them so that the pruning / clustering steps have work
    with torch.no_grad():
        batch, _, _ = run()
        sd = model.state_dict()
        last_bias = [k for k in sd if k.startswith("classifier.") and k.endswith(".bias")][-1]
        sd[last_bias] -= model(batch)["classified_edges"][-1].median()
        model.load_state_dict(sd)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        batch, probs, post = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    e = batch.edge_index.shape[1]
    print(f"{frames} frames x {cams} cameras x {per} detections: N={n} E={e}")
    print(f"graph build + MPN (L=4) + threshold/prune/cluster: {dt * 1e3:.3f} ms per batch "
          f"({e / dt / 1e6:.1f} M edges/s end to end, host planning included)")
    print(f"active edges after pruning: {int(post['pruned'].sum())}, identity clusters: {int(post['n_clusters'].item())}, "
          f"max out-flow per node: {int(post['flow_out'].max())}")
    print("random weights / random embeddings: the cluster structure is meaningless, the plumbing is what is shown")


if __name__ == "__main__":
    main()

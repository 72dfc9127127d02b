import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from gnn_cca_amd.graph_build import build_graph_batch, plan_frames
from gnn_cca_amd.postprocess import prune_and_cluster, threshold
frames, cams, per = int(sys.argv[1]), 4, 8
rng = np.random.default_rng(0)
n_g = cams*per; n = frames*n_g
id_cam = np.tile(np.repeat(np.arange(cams), per), frames)
ids = np.concatenate([rng.integers(0, per, size=n_g) for _ in range(frames)]).astype(np.int64)
xw = rng.normal(size=n); yw = rng.normal(size=n)
node = torch.randn(n, 2048, device='cuda'); reid = torch.randn(n, 256, device='cuda')
model = bench.build_model(bench.graph_net_params(), n_g).cuda().eval()
def T(f, reps=30):
    for _ in range(3): r = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3, r
t_plan, _ = T(lambda: plan_frames(id_cam, [n_g]*frames))
t_build, batch = T(lambda: build_graph_batch(xw, yw, ids, id_cam, [n_g]*frames, [80.0]*frames, node, reid))
with torch.no_grad():
    t_mpn, out = T(lambda: model(batch))
t_thr, (pr, pd) = T(lambda: threshold(out['classified_edges'][-1]))
t_post, _ = T(lambda: prune_and_cluster(batch.edge_index, pd, n, batch.node_ptr_dev, batch.edge_ptr_dev))
print(f"frames={frames} plan_frames(host) {t_plan:.3f}  build_graph_batch(total) {t_build:.3f}  mpn {t_mpn:.3f}  threshold {t_thr:.3f}  prune_cluster {t_post:.3f} ms")

"""Which earlier activity of a process decides whether GraphedForward.block(chains=3) overlaps (GPU box): run with any of the words
`single`, `big`, `eager` -- a one-forward capture, a 200-forward block, 300 eager forwards before the chains block is captured.  Round 4:
with a capture stream of its own, ANY earlier capture left the three chain streams on occupied hardware queues (26 us per forward instead
of 16); sharing the capture stream with chain 0 fixed it (profiles/r04_logs)."""
import sys, os, time, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gnn_cca_amd.inference import GraphedForward
torch.cuda.set_device(0)
model = bench.build_model(copy.deepcopy(bench.graph_net_params()), 256).to("cuda:0")
data = bench.make_data(256, 1, 1, torch.device("cuda", 0))
K = 200
def t(blk, tag):
    blk.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); blk.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
    print(f"{tag}: {sorted(ts)[3]:.2f} us per forward", flush=True)
with torch.no_grad():
    gf = GraphedForward(model, warmup=0)
    if "single" in sys.argv:
        st = gf.static_inputs(data); st.x.copy_(data.x); st.edge_index.copy_(data.edge_index); st.edge_attr.copy_(data.edge_attr); gf(st)
        print("captured single", flush=True)
    if "big" in sys.argv:
        b1 = gf.block([data] * K, adopt_inputs=True); t(b1, "big block")
    if "eager" in sys.argv:
        for _ in range(300): model(data)
        torch.cuda.synchronize()
    b3 = gf.block([data] * K, adopt_inputs=True, chains=3, depth=4); t(b3, "chains3 depth4")

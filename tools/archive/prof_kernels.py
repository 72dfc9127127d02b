"""Per-launch kernel times of one forward (events attached to every dispatch, model.forward_profiled): median over 10 forwards.
    python tools/prof_kernels.py <graphs>x<nodes>        e.g. 64x256, 512x128, 1x256
Diagnostic switches are read from the environment by the library (GNNCCA_WPS, GNNCCA_GEMM_NOPIPE, GNNCCA_GEMM_LDS_MIN, ...)."""
import copy, json, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
g, n = (int(v) for v in sys.argv[1].split("x"))
params = bench.graph_net_params(L=4)
dev = torch.device("cuda:0")
model = bench.build_model(copy.deepcopy(params), n).to(dev)
data = bench.make_data(n, g, 1, dev)
with torch.no_grad():
    for _ in range(3): model(data)
    kinds = {}
    for _ in range(10):
        _, times = model.forward_profiled(data)
        for i, (k, t) in enumerate(times): kinds.setdefault(f"{i}:{k}", []).append(t * 1e3)
print(os.environ.get("GNNCCA_WPS"), sys.argv[1], {k: round(float(np.median(v)), 1) for k, v in kinds.items()})

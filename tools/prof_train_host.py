import sys, os, cProfile, pstats, io, time
sys.argv = ["x", "64", "4", "5"]
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src = open(os.path.join(ROOT, "tools/bench_train_step.py")).read()
src = src.split("sd = {k: v.detach()")[0]   # GPU part only
src = src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT))
g = {"__name__": "__main__", "__file__": os.path.join(ROOT, "tools/bench_train_step.py")}
exec(compile(src, "bench_train_step", "exec"), g)
import torch
step = g["step"]
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(50): step()
t_issue = (time.perf_counter() - t0) / 50
torch.cuda.synchronize()
pr.disable()
print("host issue ms/step", t_issue * 1e3, " wall ms/step", (time.perf_counter() - t0) / 50 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])

#!/usr/bin/env python3
"""Throughput of the headline workload (one dense 256-node graph per forward) with S independent forwards in flight:
ONE module, S streams, each stream replaying its own captured HIP graph.  usage (GPU box): python3 tools/bench_streams.py [S]"""
import sys, os, time, copy
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch, bench
params = bench.graph_net_params()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = bench.build_model(copy.deepcopy(params), 256).cuda()  # ONE module: a workspace per stream, shared weights
models = [model] * S
datas = [bench.make_data(256, 1, 1 + i, "cuda") for i in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
graphs, outs = [], []
with torch.no_grad():
    for m, d, st in zip(models, datas, streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(3): m(d)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            o = m(d)
        graphs.append(g); outs.append(o)
    E = datas[0].edge_index.shape[1]
    def run(n):
        for _ in range(n):
            for g, st in zip(graphs, streams):
                with torch.cuda.stream(st):
                    g.replay()
    run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 2000; run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{S} streams: {dt / (K * S) * 1e6:.2f} us per forward, {E * K * S / dt / 1e9:.2f} G edges/s", flush=True)

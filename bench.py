#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X message-passing path on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nodes 256] [--graphs 1] [--L 4] [--mode auto|eager|graph|graphk|graphs]

A "step" is one MOTMPNet.forward (encoder + L message-passing steps + the classifier on the last 3 steps, eval
mode, no grad) over one batch of synthetic input already resident in HBM: `--graphs` independent fully-connected
`--nodes`-node graphs as one disjoint union (what Batch.from_data_list hands the reference, inference.py:279).
Default = ONE 256-node dense graph (65 280 edges), L = 4: the configuration BASELINE.json's metric is quoted on.

N > 1, one rank process per GPU: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment) or -- when WORLD_SIZE is unset -- by this script itself: the parent touches no GPU API, starts N fresh
children of itself with those variables set, waits for them and exits non-zero if any of them does.  Every rank runs
the same per-GPU workload on its own graphs (weak scaling; independent graphs, no cross-GPU edges, no data-path
collective); rank 0 packs the weights and broadcasts the blob over RCCL.  value = edges processed by all ranks /
max-over-ranks time.  The line also carries `config4_sharded`: BASELINE config 4 (512 independent dense128 graphs)
sharded 512/N per rank through gnn_cca_amd.sharding.forward_sharded, timed the same way (total work fixed: strong).

Timing: >= 10 blocks of exactly `--steps` steps, each block bracketed by barrier + synchronize on both sides and reduced
with MAX over ranks; the MEDIAN block is reported (ms_per_step = median block / steps), so a short `--steps` run is not
one sub-millisecond sample.  How the K steps of a block are issued is `--mode`, each form through the package API a caller of the
reference's per-frame loop (inference.py:173-283) would use: K eager calls of MOTMPNet.forward; K replays of
gnn_cca_amd.inference.GraphedForward's one-forward HIP graph; ONE replay of GraphedForward.block -- the K forwards, every launch of
each, nothing cached or skipped, captured back to back in one HIP graph; or the same K forwards as HIP graphs of four frames round
robin on `--streams` streams (GraphedForward.block(chains=S): independent frames in flight).  `auto` times every form with the same
block protocol and lists each in `config.ms_per_step_by_mode`; `value` / `ms_per_step` / `config.gpu_over_cpu` are the fastest form that
runs ONE forward at a time (eager, graph, graph_block: SURVEY 8d's E / t_forward, the definition of rounds 1-3), named in
`config.mode`; the frames-in-flight figure lives in `pipelined` only (`--mode graphs` selects it by name for experiments).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel mpn_step_kernel, per-launch
algorithmic bytes / HIP-event duration) and `cpu_baseline` (the reference-shaped torch CPU
restatement in oracle/, timed on this box's host cores; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def graph_net_params(L=4, n_cls=3, agg="sum", cls_bn=True):
    """GRAPH_NET_PARAMS of the reference's shipped inference config (config_inference.yaml:76-163)."""
    return {
        "node_agg_fn": agg, "num_enc_steps": L, "num_class_steps": n_cls,
        "reattach_initial_nodes": False, "reattach_initial_edges": False,
        "encoder_feats_dict": {
            "edges": {"edge_in_dim": 4, "edge_fc_dims": [], "edge_out_dim": 6},
            "nodes": {"resnet50": {"node_in_dim": 2048, "node_fc_dims": [128], "node_out_dim": 32,
                                   "dropout_p": 0, "use_batchnorm": False}},
        },
        "edge_model_feats_dict": {"fc_dims": [6], "dropout_p": 0, "use_batchnorm": False},
        "node_model_feats_dict": {"fc_dims": [32], "dropout_p": 0, "use_batchnorm": False},
        "classifier_feats_dict": {"edge_in_dim": 6, "edge_fc_dims": [4], "edge_out_dim": 1, "dropout_p": 0,
                                  "use_batchnorm": cls_bn},
    }


def build_model(params, n_nodes, seed=0):
    """Random-init weights of the reference architecture (no checkpoint exists offline), conditioned as SURVEY.md
    7.3 prescribes: node-MLP weight and bias x 1/(N-1) so 'sum' aggregation keeps activations O(1)."""
    from gnn_cca_amd import MOTMPNet
    torch.manual_seed(seed)
    m = MOTMPNet(params, None, "resnet50")
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.weight.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.num_features, generator=g))
        for p in m.MPNet.node_model.node_mlp.parameters():
            p.mul_(1.0 / max(n_nodes - 1, 1))
    return m.eval()


def dense_union(n_nodes, n_graphs, device):
    """edge_index of `n_graphs` disjoint fully-connected directed graphs, i-major (row sorted)."""
    i = torch.arange(n_nodes, device=device).repeat_interleave(n_nodes)
    j = torch.arange(n_nodes, device=device).repeat(n_nodes)
    keep = i != j
    ei = torch.stack([i[keep], j[keep]])  # [2, n(n-1)]
    offs = (torch.arange(n_graphs, device=device) * n_nodes).view(-1, 1, 1)
    return (ei.unsqueeze(0) + offs).permute(1, 0, 2).reshape(2, -1).contiguous()


class Data:
    pass


def make_data(n_nodes, n_graphs, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n_graphs * n_nodes, 2048, generator=g)
    x = torch.nn.functional.normalize(x, p=2, dim=0)  # inference.py:189-190
    d = Data()
    d.x = x.to(device)
    d.edge_index = dense_union(n_nodes, n_graphs, device)
    d.edge_attr = torch.rand(d.edge_index.shape[1], 4, generator=g).to(device)
    return d


def step_algorithmic_bytes(E, L, n_cls, msg_only=True, e_bytes=24):
    """Algorithmic HBM bytes of the mpn_step_kernel launches of one forward (fp32):
    read e (24 B; step 1 reads edge_attr, 16 B) + col32 (4 B) + write e' (24 B, not on the last step) + logit (4 B)."""
    first_cls = L - n_cls + 1
    per_launch = []
    for s in range(1, L + 1):
        b = (16 if s == 1 else e_bytes) + 4 + (e_bytes if s < L else 0) + (4 if s >= first_cls else 0)
        if msg_only and s == L:
            continue
        per_launch.append(b * E)
    return per_launch


def forward_algorithmic_bytes(N, E, L=4, n_cls=3, node_in=2048, e_bytes=24):
    """Algorithmic HBM bytes of one whole forward (DESIGN.md section 4's per-kernel figures added up): every step launch (edge state in / out,
    col32, logits), the plan (16 B of int64 ids read + 4 B of col32 written per edge), the encoder's x stream + 1 MB of first-layer weights."""
    return float(sum(step_algorithmic_bytes(E, L, n_cls, msg_only=False, e_bytes=e_bytes)) + 20 * E + N * node_in * 4 + (1 << 20))


def leg_fractions(kernels_us, N, E, ms_per_step, msg_bytes_mean):
    """HBM fractions of a batch forward from its per-kernel times: encoder (x stream + weights over the enc_gemm launch), mean message step,
    whole forward (algorithmic bytes of the forward over its wall time per step)."""
    out = {}
    if kernels_us.get("enc_gemm"):
        out["enc_frac"] = (N * 2048 * 4 + (1 << 20)) / (kernels_us["enc_gemm"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
    if kernels_us.get("step"):
        out["step_frac"] = msg_bytes_mean / (kernels_us["step"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
    if ms_per_step:
        out["forward_frac"] = forward_algorithmic_bytes(N, E) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS
    return out


def scale_probe(params, device, args, graphs=64):
    """The SAME step kernel on a batch of `graphs` graphs of the headline size, where the edge state (200 MB) no longer
    fits any cache: the HBM-relevant operating point of the dominant kernel, measured live (HIP events attached to each
    launch, `forward_profiled`), reported next to the latency-bound single-graph figure.  Not part of `value`."""
    import copy
    model = build_model(copy.deepcopy(params), args.nodes).to(device)
    model.edge_state_dtype = "bf16" if args.edge_state == "bf16" else "fp32"
    data = make_data(args.nodes, graphs, 1, device)
    E = data.edge_index.shape[1]
    ms = []
    with torch.no_grad():
        for _ in range(3):
            model(data)
        for _ in range(8):
            _, times = model.forward_profiled(data)
            ms += [t for kind, t in times if kind == "step"]
    per_launch = step_algorithmic_bytes(E, args.L, 3, e_bytes=12 if args.edge_state == "bf16" else 24)
    alg = float(np.mean(per_launch))
    step_ms = float(np.mean(ms))
    achieved = alg / (step_ms * 1e-3) / 1e9
    return {"workload": f"{graphs} x dense{args.nodes} graphs in one forward (E={E})", "bound": "hbm", "achieved": achieved,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "avg_launch_us": step_ms * 1e3,
            "algorithmic_bytes_per_launch": alg, "kernel": "mpn_step_pipe_kernel (the step kernel of graphs / batches beyond 512 nodes; the headline graph runs mpn_step_fast_kernel)"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(params, model, n_nodes, n_graphs, budget_s=20.0):
    """The reference-shaped CPU path (oracle.TorchOracle: index/cat/addmm/relu/index_add_, the torch CPU kernels the
    reference itself runs) on a bounded sample of the same workload."""
    import copy

    from oracle.mpn_oracle import TorchOracle
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    orc = TorchOracle(copy.deepcopy(params), "resnet50", sd)
    g_sample = min(n_graphs, 4)
    d = make_data(n_nodes, g_sample, 1, "cpu")
    E = d.edge_index.shape[1]
    # torch's CPU kernels do not scale to every core of a big host on this small problem: time a few thread counts
    # inside the budget and report the FASTEST (most favourable to the CPU), naming the others in `sample`.
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (1, 8, 16, 32, 64) if c <= ncpu} | ({ncpu} if ncpu <= 64 else set()))   # k = 1 and many
    results = {}
    for c in cands:
        torch.set_num_threads(c)
        for _ in range(2):
            orc.forward(d.x, d.edge_index, d.edge_attr)
        times = []
        t_end = time.perf_counter() + budget_s / len(cands)
        while len(times) < 30 and (time.perf_counter() < t_end or len(times) < 3):
            t0 = time.perf_counter()
            orc.forward(d.x, d.edge_index, d.edge_attr)
            times.append(time.perf_counter() - t0)
        results[c] = (float(np.median(times)), len(times))
    best = min(results, key=lambda c: results[c][0])
    med, n_runs = results[best]
    others = ", ".join(f"{c}thr {results[c][0] * 1e3:.1f}ms" for c in cands)
    # the same sample through the HIP path, checked against the oracle's logits (the parity metric next to the
    # throughput): max |logit_gpu - logit_cpu| per classified step and the largest deviation relative to max |logit|
    parity = None
    try:
        ref = orc.forward(d.x, d.edge_index, d.edge_attr)
        dev = next(model.parameters()).device
        dg = Data()
        dg.x, dg.edge_index, dg.edge_attr = d.x.to(dev), d.edge_index.to(dev), d.edge_attr.to(dev)
        with torch.no_grad():
            got = model(dg)["classified_edges"]
        errs = [float((g.cpu() - torch.as_tensor(r)).abs().max()) for g, r in zip(got, ref)]
        scale = max(float(torch.as_tensor(r).abs().max()) for r in ref)
        parity = {"max_abs_err": errs, "max_rel_err": max(errs) / max(scale, 1e-30), "max_abs_logit": scale, "tolerance_abs": 1e-4,
                  "ok": bool(max(errs) <= 1e-4), "against": "oracle.TorchOracle (fp32, CPU) on the cpu_baseline sample"}
    except Exception as exc:  # noqa: BLE001
        parity = {"error": f"{type(exc).__name__}: {exc}"}
    return {"value": E / med, "unit": "edges/s", "cores": best, "kind": "port", "parity": parity,
            "sample": f"{n_runs} fwd of {g_sample} x dense{n_nodes}, median {med * 1e3:.2f} ms @ {best} thr, oracle.TorchOracle fp32",
            "ms_per_forward": med * 1e3, "host": f"{_cpu_model()}, {ncpu} hw threads"[:100], "medians_by_threads": others[:100]}


class LazyDenseGraphs:
    """512 (or any number of) independent dense graphs as a sequence that materialises only the slice asked for, so each
    rank builds just its own share (gnn_cca_amd.sharding.shard_batch slices it by the rank's range).  Graph g is the
    same on every rank and for every world size: seeded by g."""

    def __init__(self, n_graphs, n_nodes, device):
        self.n_graphs, self.n_nodes, self.device = n_graphs, n_nodes, device
        self._ei = dense_union(n_nodes, 1, device)

    def __len__(self):
        return self.n_graphs

    def __getitem__(self, sl):
        out = []
        for g in range(*sl.indices(self.n_graphs)):
            gen = torch.Generator(device=self.device).manual_seed(7000 + g)
            x = torch.randn(self.n_nodes, 2048, generator=gen, device=self.device)
            x = torch.nn.functional.normalize(x, p=2, dim=0)
            ea = torch.rand(self._ei.shape[1], 4, generator=gen, device=self.device)
            out.append((x, self._ei, ea))
        return out


def timed_blocks(run, steps, warmup, dist, device, backend, min_blocks=10, min_total_s=0.25, max_blocks=200, run_block=None):
    """W untimed warm-up steps, then blocks of EXACTLY `steps` steps; every block is bracketed by barrier +
    torch.cuda.synchronize() on both sides and its time is the MAX over ranks (of the larger of host wall clock and the
    HIP-event span on the launch stream).  Returns the sorted block times; the caller reports the median.  The block
    count is the same on every rank because every block time is all-reduced."""
    if run is None:
        run_block()                   # a block form warms up with one whole (untimed) block
    else:
        for _ in range(warmup):
            run()
    blocks, out = [], None
    while len(blocks) < min_blocks or (sum(blocks) < min_total_s and len(blocks) < max_blocks):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        if run_block is not None:      # the K steps of a block as ONE HIP-graph launch (K forwards captured back to back)
            out = run_block()
        else:
            for _ in range(steps):
                out = run()
        ev1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0   # this rank's K steps, drained; the MAX over ranks below is the job's time
        if dist:
            dist.barrier()                # closing bracket (its own latency is not part of any rank's K steps)
        torch.cuda.synchronize()
        t = max(wall, ev0.elapsed_time(ev1) / 1e3)
        if dist:
            tt = torch.tensor([t], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        blocks.append(t)
    return sorted(blocks), out


FORM_ORDER = ["eager", "graph", "graph_block", "graph_block_chains"]


def agree_forms(have, dist, device, backend):
    """Multi-rank runs: the issue forms EVERY rank holds.  Each rank captures its HIP graphs under its own try / except, so a capture that fails
    on one rank only would leave that rank skipping a form whose timed blocks the others enter -- the barriers / all-reduces inside
    `timed_blocks` would then pair up across DIFFERENT forms and the job would hang.  One all-reduce (MIN) over a 0 / 1
    vector in FORM_ORDER before anything is timed: a form missing anywhere is dropped everywhere.  Returns the agreed names in FORM_ORDER."""
    mine = [1 if k in have else 0 for k in FORM_ORDER]
    if dist is not None:
        t = torch.tensor(mine, dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        mine = [int(v) for v in t.tolist()]
    return [k for k, ok in zip(FORM_ORDER, mine) if ok]


def agree_flag(flag, dist, device, backend):
    """True only if `flag` is true on every rank (same all-reduce, one word): eligibility decisions that gate a COLLECTIVE region."""
    if dist is None:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def flat_evidence(res):
    """The scalars the line's claims rest on, copied into `config` / `roofline` under short flat keys: the driver's record
    keeps the scalar entries of those two objects and only the NAMES of the nested legs, so rows (d), (e), N1-N3 could not be checked from
    BENCH_rNN.json alone.  Every value here is a copy of a figure of the nested objects of the same line, nothing is computed twice."""
    c, r = res["config"], res["roofline"]

    def put(dst, key, val):
        if val is not None and not isinstance(val, (dict, list)):
            dst[key] = val

    for k, v in (res["config"].get("ms_per_step_by_mode") or {}).items():
        put(c, f"ms_{k}", v)
    sh, c4 = res.get("config4_share"), res.get("config4_sharded")
    if c4:
        put(c, "cfg4_graphs_per_rank", c4.get("graphs_per_rank"))
        put(c, "cfg4_ms", c4.get("ms_per_step"))
        put(c, "cfg4_ms_eager", c4.get("ms_per_step_eager"))
        put(c, "cfg4_ms_graph", c4.get("ms_per_step_graph"))
        put(c, "cfg4_graph_equals_eager", c4.get("graph_equals_eager_bitwise"))
        put(c, "cfg4_edges_per_s", c4.get("value"))
        rr = c4.get("roofline_rank0") or {}
        put(r, "cfg4_step_frac", rr.get("frac"))
        put(r, "cfg4_step_us", rr.get("avg_launch_us"))
        put(r, "cfg4_enc_frac", c4.get("enc_frac_rank0"))
        put(r, "cfg4_enc_us", (c4.get("kernels_us_rank0") or {}).get("enc_gemm"))
        put(r, "cfg4_forward_frac", c4.get("forward_frac_rank0"))
    if sh:
        put(c, "share_ms", sh.get("ms_per_step"))
        put(c, "union_ms", sh.get("union_ms_per_step"))
        put(c, "projected_8gpu_speedup", sh.get("projected_8gpu_speedup"))
        put(c, "share_ms_eager", sh.get("ms_per_step_eager"))
        put(c, "share_ms_graph", sh.get("ms_per_step_graph"))
        put(c, "projected_8gpu_speedup_eager", sh.get("projected_8gpu_speedup_eager"))
        put(c, "projected_8gpu_is", "a PROJECTION: union ms / share ms on ONE GPU, not an 8-GPU measurement")
        put(r, "share_enc_frac", sh.get("enc_frac"))
        put(r, "share_step_frac", sh.get("step_frac"))
        put(r, "share_forward_frac", sh.get("forward_frac"))
        for k, v in (sh.get("kernels_us") or {}).items():
            put(r, f"share_{k}_us", v)
    ras = res.get("roofline_at_scale")
    if ras:
        put(r, "at_scale_frac", ras.get("frac"))
        put(r, "at_scale_step_us", ras.get("avg_launch_us"))
    tp = res.get("terrace_pipeline")
    if tp and "error" not in tp:
        put(c, "terrace_ms_per_batch", tp.get("ms_per_batch"))
        put(c, "terrace_frames_per_s", tp.get("frames_per_s"))
        fin = tp.get("with_rounding_and_splitting") or {}
        put(c, "terrace_final_ms_per_batch", fin.get("ms_per_batch"))
        put(c, "terrace_final_frames_per_s", fin.get("frames_per_s"))
        put(c, "terrace_final_overlapped_ms_per_batch", fin.get("overlapped_ms_per_batch"))
        put(c, "terrace_final_overlapped_frames_per_s", fin.get("overlapped_frames_per_s"))
        put(c, "terrace_final_over_chain", tp.get("final_over_chain"))
        put(c, "terrace_flagged_per_batch", fin.get("frames_through_the_host_heuristics_per_batch"))
        put(c, "terrace_parity_ok", (tp.get("parity") or {}).get("ok"))
    tr = res.get("train_step")
    if tr and "error" not in tr:
        put(c, "train_ms_per_iteration", tr.get("ms_per_iteration"))
        put(c, "train_grad_err", (tr.get("parity") or {}).get("grad_max_abs_err"))
        put(c, "train_parity_ok", (tr.get("parity") or {}).get("ok"))
    short = {"config2_dense64_L4_fp32": "cfg2", "config3_dense256_L4_bf16_state": "cfg3bf16", "config5_dense1024_L8_fp32": "cfg5",
             "config5_dense1024_L8_bf16_state": "cfg5bf16"}
    for key, leg in (res.get("configs") or {}).items():
        if "error" in leg:
            put(c, f"{short.get(key, key)}_error", str(leg["error"])[:90])
            continue
        k = short.get(key, key)
        put(c, f"{k}_ms", leg.get("ms_per_step"))
        put(c, f"{k}_err", max((leg.get("parity") or {}).get("max_abs_err") or [float("nan")]))
        put(r, f"{k}_step_frac", (leg.get("roofline") or {}).get("frac"))
    par = res.get("parity")
    if par:
        put(c, "headline_err", max(par.get("max_abs_err") or [float("nan")]))
        put(c, "headline_parity_ok", par.get("ok"))
    dr = res.get("dynamic_range")
    if dr and "error" not in dr:
        for k, v in dr.items():
            put(c, f"dynrange_{k}", v)
    # the record truncates strings near 120 characters: every string of `config` / `roofline` / `cpu_baseline` stays under 100, the long form
    # moves to `notes` (a nested object: in the line, not in the record)
    notes = res.setdefault("notes", {})
    for name in ("config", "roofline", "cpu_baseline"):
        d_ = res.get(name) or {}
        for k, v in list(d_.items()):
            if isinstance(v, str) and len(v) > 100:
                notes[f"{name}.{k}"] = v
                d_[k] = v[:96] + " ..."
    return res


def state_hash(model):
    """64-bit hash of the module's whole state_dict (bytes of every tensor in key order), computed on the host."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return int.from_bytes(h.digest(), "little", signed=True)


def broadcast_and_verify(model, dist, device, backend, rank, world):
    """The ONE collective of the path: rank 0's state_dict to every rank (RCCL over xGMI for backend nccl), timed; then every rank's
    state_dict hash is all-gathered and compared with rank 0's -- a rank that kept its own weights fails the job loudly instead
    of producing plausible numbers.  Returns (broadcast milliseconds on this rank, this rank's hash)."""
    from gnn_cca_amd.sharding import broadcast_weights
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    broadcast_weights(model, src=0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    mine = state_hash(model)
    dev = device if backend == "nccl" else "cpu"
    hashes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(hashes, torch.tensor([mine], dtype=torch.int64, device=dev))
    got = [int(h.item()) for h in hashes]
    if any(h != got[0] for h in got):
        raise SystemExit(f"bench.py: rank {rank}: state_dict hashes differ after the weight broadcast: {got}")
    return ms, mine


def launch_ranks(args):
    """--gpus N > 1 without WORLD_SIZE: this process becomes the launcher.  It never touches a GPU API (no
    torch.cuda call at all); it starts N FRESH copies of this script -- one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set -- waits for them, and exits non-zero if any of them fails.  Rank 0's stdout (the one
    JSON line) is this process's stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, deadline = 0, time.time() + args.launch_timeout
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if alive and (rc != 0 or time.time() > deadline):   # one rank failed (or the job hung): stop the others by PID
            for p in alive:
                p.terminate()
            t_kill = time.time() + 10
            for p in alive:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
    if rc != 0:
        print(f"[bench] a rank process failed (exit code {rc})", file=sys.stderr)
    sys.exit(rc if 0 <= rc < 256 else 1)


def cpu_baseline_fused(params, model, n_nodes, n_graphs, budget_s=8.0):
    """SURVEY.md 8(d) flavour (ii), for information: the fused split-weight C / OpenMP restatement (oracle/mpn_oracle_c.c,
    no [E,70] / [E,38] concatenations) on the same sample, all host cores OpenMP gives it."""
    import copy

    from oracle.c_oracle import COracle
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    orc = COracle(copy.deepcopy(params), "resnet50", sd)
    g_sample = min(n_graphs, 4)
    d = make_data(n_nodes, g_sample, 1, "cpu")
    x, ei, ea = d.x.numpy(), d.edge_index.numpy(), d.edge_attr.numpy()
    E = ei.shape[1]
    for _ in range(2):
        orc.forward(x, ei, ea)
    times, t_end = [], time.perf_counter() + budget_s
    while len(times) < 30 and (time.perf_counter() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        orc.forward(x, ei, ea)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": E / med, "unit": "edges/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": f"{len(times)} forwards of {g_sample} x dense{n_nodes} (E={E}), median {med * 1e3:.2f} ms, OpenMP default "
                      f"threads on a {os.cpu_count()}-thread host; fused split-weight C restatement (oracle/mpn_oracle_c.c), fp32"}


# ---- the reference's actual inference workload: real frame shapes of EPFL-Terrace through rows N1 + MPN + N2 ------------------------
def terrace_frames(batch, n_batches, seed=0):
    """Batches of `batch` consecutive valid frames of the reference's own sequence (tests/golden/terrace_topology.npz: camera and
    person id of every detection, derived from datasets/EPFL-Terrace/*/gt/gt.txt by tests/golden/make_terrace_topology.py --
    4816 valid frames, 19.1 detections / 343 edges per frame on average, at most 34 / 866).  Images and ReID weights are not in the
    repository, so positions and embeddings are SYNTHETIC: a ground-plane position and a 256-d / 2048-d appearance vector per
    person, plus detection noise.  Returns a list of dicts of host arrays (what libs/datasets.py + the CNN would hand over)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "terrace_topology.npz"))
    ptr, cam, person = z["node_ptr"], z["cam"].astype(np.int64), z["person"].astype(np.int64)
    n_frames = len(ptr) - 1
    rng = np.random.default_rng(seed)
    n_person = int(person.max()) + 1
    app_reid = rng.standard_normal((n_person, 256)).astype(np.float32)
    app_node = rng.standard_normal((n_person, 2048)).astype(np.float32)
    pos = rng.uniform(-8.0, 8.0, size=(n_person, 2))
    out = []
    for b in range(n_batches):
        f0 = (b * max((n_frames - batch) // max(n_batches, 1), 1)) % max(n_frames - batch, 1)   # batches spread over the whole sequence
        lo, hi = int(ptr[f0]), int(ptr[f0 + batch])
        ids, cams = person[lo:hi], cam[lo:hi]
        n = hi - lo
        out.append({"sizes": np.diff(ptr[f0:f0 + batch + 1]).astype(np.int64), "id_cam": cams, "ids": ids,
                    "xw": pos[ids, 0] + rng.normal(0, 0.3, n), "yw": pos[ids, 1] + rng.normal(0, 0.3, n),
                    "max_dist": np.full(batch, 80.0),
                    "node": (app_node[ids] + 0.5 * rng.standard_normal((n, 2048))).astype(np.float32),
                    "reid": (app_reid[ids] + 0.5 * rng.standard_normal((n, 256))).astype(np.float32)})
    return out


def nat_threads(pipe):
    from gnn_cca_amd import _native as nat
    return nat.lib().gnncca_post_pool_threads(pipe._pool_handle())


def terrace_leg(device, args, batch=64, n_batches=16, cpu_budget_s=12.0):
    """`terrace_pipeline`: batches of 64 real-shaped frames (train BATCH_SIZE, config_training.yaml:53-54) through
    build_graph_batch -> MOTMPNet -> threshold -> prune_and_cluster (inference.py:189-345 without the bridge heuristics), every
    batch a different set of frames (no shape repeats, so nothing is graph-captured: eager calls); the CPU oracle chain
    (oracle.graph_oracle + oracle.TorchOracle + oracle.post_oracle) is the parity check and the baseline."""
    import copy

    from gnn_cca_amd.graph_build import build_graph_batch
    from gnn_cca_amd.postprocess import prune_and_cluster, threshold
    from oracle import graph_oracle as go
    from oracle import post_oracle as po
    from oracle.mpn_oracle import TorchOracle
    params = graph_net_params(L=4)
    frames = terrace_frames(batch, n_batches)
    model = build_model(copy.deepcopy(params), 20, seed=0).to(device)    # out-degrees ~19: node-MLP conditioned with 1/19
    dev_in = [(torch.from_numpy(f["node"]).to(device), torch.from_numpy(f["reid"]).to(device)) for f in frames]

    from gnn_cca_amd.pipeline import FramePipeline
    pipe = FramePipeline(model)

    def run(i):
        # one native call per batch (gnncca_frames_forward): the launches of build_graph_batch -> model(b) -> threshold -> prune_and_cluster,
        # bit for bit their results (tests/test_gpu_pipeline.py), without the Python between them
        f, (node, reid) = frames[i], dev_in[i]
        r = pipe(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
        return r.batch, r.outputs, r.preds, {"pruned": r.pruned, "labels": r.labels, "n_clusters": r.n_clusters, "_keep": r}

    # random weights put every logit on one side of 0: centre them on batch 0 so that pruning / clustering have work to do
    with torch.no_grad():
        b0, out0, _, _ = run(0)
        sd = model.state_dict()
        last_bias = [k for k in sd if k.startswith("classifier.") and k.endswith(".bias")][-1]
        sd[last_bias] -= out0["classified_edges"][-1].median()
        model.load_state_dict(sd)
    for i in range(min(3, n_batches)):
        run(i)
    torch.cuda.synchronize()
    stage = {"graph_build": 0.0, "mpn": 0.0, "post": 0.0}
    t0 = time.perf_counter()
    reps = 3
    edges = nodes = 0
    for _ in range(reps):
        for i in range(n_batches):
            b, _, _, _ = run(i)
            edges += b.edge_index.shape[1]
            nodes += b.x.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same batches with the reference's rounding / splitting heuristics (config_inference.yaml:6-8 all True): FrameResult.final() reads the
    # device chain's trigger words (one synchronisation) and runs the flagged frames through the native host implementation
    t0 = time.perf_counter()
    flagged = 0
    for i in range(n_batches):
        _, _, _, post = run(i)
        flagged += len(post["_keep"].final()["frames_finalized"])
    torch.cuda.synchronize()
    dt_final = time.perf_counter() - t0
    # ... and OVERLAPPED (round 6): FrameResult.final_async() hands batch k to the pipeline's pool of host threads behind one D2H copy enqueued
    # behind the chain -- no synchronisation -- while this loop enqueues batch k + 1's chain; a batch is collected four batches later.  Every
    # batch still ends with its FINAL host-side predictions / labels (what the reference's loop has after inference.py:345).
    depth, pend, flagged_o, done_o = 4, [], 0, 0
    for i in range(min(4, n_batches)):      # pool threads, pinned buffers, the copy stream: created on first use
        run(i)[3]["_keep"].final_async().result()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(n_batches):
            pend.append(run(i)[3]["_keep"].final_async())
            if len(pend) > depth:
                flagged_o += len(pend.pop(0).result()["frames_finalized"])
                done_o += 1
    while pend:
        flagged_o += len(pend.pop(0).result()["frames_finalized"])
        done_o += 1
    torch.cuda.synchronize()
    dt_over = time.perf_counter() - t0
    for i in range(n_batches):   # stage split (synchronised between stages: for the split only, not part of the figure above)
        f, (node, reid) = frames[i], dev_in[i]
        torch.cuda.synchronize(); t = time.perf_counter()
        b = build_graph_batch(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], node, reid)
        torch.cuda.synchronize(); stage["graph_build"] += time.perf_counter() - t; t = time.perf_counter()
        with torch.no_grad():
            out = model(b)
        torch.cuda.synchronize(); stage["mpn"] += time.perf_counter() - t; t = time.perf_counter()
        probs, preds = threshold(out["classified_edges"][-1])
        prune_and_cluster(b.edge_index, preds, b.x.shape[0], b.node_ptr_dev, b.edge_ptr_dev)
        torch.cuda.synchronize(); stage["post"] += time.perf_counter() - t
    # ---- parity + CPU baseline: the oracle chain on the same batches --------------------------------------------------------------
    sdn = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    orc = TorchOracle(copy.deepcopy(params), "resnet50", sdn)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cpu_t, cpu_edges, cpu_frames, checks = 0.0, 0, 0, []
    check = {0, max(range(n_batches), key=lambda q: len(frames[q]["ids"]))}   # the first and the largest batch
    for i in range(n_batches):
        f = frames[i]
        tc = time.perf_counter()
        reid_n, node_n = go.normalize_columns(f["reid"]), go.normalize_columns(f["node"])
        ei, ea, _ = go.build(f["xw"], f["yw"], f["ids"], f["id_cam"], f["sizes"], f["max_dist"], reid_n)
        ref = orc.forward(node_n, ei, ea)
        r_probs, r_preds = po.threshold(np.asarray(ref[-1]).reshape(-1))
        r_pruned = po.prune(ei, r_preds)
        n = node_n.shape[0]
        r_lab, r_k = po.clusters(ei, r_pruned, n)
        cpu_t += time.perf_counter() - tc
        cpu_edges += ei.shape[1]
        cpu_frames += batch
        if i in check:
            b, out, preds, post = run(i)
            torch.cuda.synchronize()
            g_preds = preds.cpu().numpy()
            r_logit = np.asarray(ref[-1]).reshape(-1)
            firm = np.abs(r_logit) > 1e-4     # an edge whose logit sits on the threshold may legitimately flip
            # the post-processing is checked on the GPU's own predictions (a flipped borderline edge changes the clusters for both)
            p_pruned = po.prune(ei, g_preds)
            p_lab, p_k = po.clusters(ei, p_pruned, n)
            one = {"batch": i, "detections": int(n), "edges": int(ei.shape[1]),
                   "edge_index_equal": bool(np.array_equal(b.edge_index.cpu().numpy(), ei)),
                   "edge_attr_max_abs_err": float(np.abs(b.edge_attr.cpu().numpy() - ea).max()),
                   "logit_max_abs_err": max(float(np.abs(o.view(-1).cpu().numpy() - np.asarray(r).reshape(-1)).max())
                                            for o, r in zip(out["classified_edges"], ref)),
                   "prediction_flips_on_firm_logits": int((g_preds != r_preds)[firm].sum()),
                   "prediction_flips_within_1e-4_of_the_threshold": int((g_preds != r_preds)[~firm].sum()),
                   "pruned_equal": bool(np.array_equal(post["pruned"].cpu().numpy(), p_pruned)),
                   "clusters_equal": bool(int(post["n_clusters"].item()) == p_k and po.same_partition(post["labels"].cpu().numpy(), p_lab)),
                   "clusters": int(post["n_clusters"].item()), "active_edges": int(post["pruned"].sum().item())}
            # ... and the FINAL result (ROUNDING / PRUNING / SPLITTING): every frame of the batch through the oracle's restatement of the
            # reference's heuristics, on the GPU's own probabilities (they compare probabilities for order and equality)
            fin = post["_keep"].final()
            f_pred, f_lab, g_probs = fin["predictions"].cpu().numpy(), fin["labels"].cpu().numpy(), post["_keep"].probs.cpu().numpy()
            f_ok, f_clusters = True, 0
            for q in range(batch):
                v0, v1, k0, k1 = b.node_ptr[q], b.node_ptr[q + 1], b.edge_ptr[q], b.edge_ptr[q + 1]
                _, want, ids, kq = po.finalize(ei[:, k0:k1] - v0, None, v1 - v0, probs=g_probs[k0:k1])
                f_ok = f_ok and np.array_equal(f_pred[k0:k1], want) and po.same_partition(f_lab[v0:v1], ids)
                f_clusters += kq
            one["final_equal_to_reference_heuristics"] = bool(f_ok and int(fin["n_clusters"].item()) == f_clusters)
            one["final_clusters"], one["frames_through_the_host_heuristics"] = f_clusters, len(fin["frames_finalized"])
            one["ok"] = bool(one["edge_index_equal"] and one["edge_attr_max_abs_err"] <= 1e-5 and one["logit_max_abs_err"] <= 1e-4 and
                             one["prediction_flips_on_firm_logits"] == 0 and one["pruned_equal"] and one["clusters_equal"] and
                             one["final_equal_to_reference_heuristics"])
            checks.append(one)
        if cpu_t > cpu_budget_s:
            break
    n_done = reps * n_batches
    return {"workload": f"{n_batches} batches of {batch} consecutive valid EPFL-Terrace frames (real per-frame camera / identity "
                        f"structure, synthetic positions and embeddings): {nodes // n_done} detections, {edges // n_done} edges per batch",
            "pipeline": "gnn_cca_amd.pipeline.FramePipeline: graph_build.build_graph_batch -> MOTMPNet.forward (L=4) -> postprocess.threshold -> "
                        "postprocess.prune_and_cluster (+ trigger words) as ONE native call per batch (gnncca_frames_forward), host planning "
                        "and H2D of the per-detection arrays included; `stage_ms_per_batch_synchronised` times the separate functions",
            "ms_per_batch": dt / n_done * 1e3, "frames_per_s": batch * n_done / dt, "edges_per_s": edges / dt,
            # the pipeline's rate WITH the reference's final partitions (final_async, overlapped) and its cost over the chain alone (target <= 1.5)
            "final_frames_per_s": batch * done_o / dt_over, "final_over_chain": (dt_over / done_o) / (dt / n_done),
            "with_rounding_and_splitting": {"ms_per_batch": dt_final / n_batches * 1e3, "frames_per_s": batch * n_batches / dt_final,
                                            "frames_through_the_host_heuristics_per_batch": flagged / n_batches,
                                            "overlapped_ms_per_batch": dt_over / done_o * 1e3, "overlapped_frames_per_s": batch * done_o / dt_over,
                                            "overlapped_flagged_per_batch": flagged_o / done_o, "overlapped_depth": depth,
                                            "overlapped_host_threads": int(nat_threads(pipe)),
                                            "note": "ms_per_batch: FrameResult.final() per batch, one after the other (waits for the batch, host pass, results "
                                                    "uploaded again); overlapped_*: FrameResult.final_async(), batch k's host pass (pool of host threads, "
                                                    "csrc/post_host.cpp) overlaps batch k + 1's GPU chain, results collected four batches later.  The synthetic "
                                                    "model's predictions are near-random, so MOST frames raise a trigger here; a trained model's rarely do"},
            "stage_ms_per_batch_synchronised": {k: v / n_batches * 1e3 for k, v in stage.items()},
            "parity": {"ok": bool(checks) and all(c["ok"] for c in checks), "tolerance_abs": 1e-4, "batches": checks,
                       "against": "oracle.graph_oracle + oracle.TorchOracle on the same inputs; oracle.post_oracle on the GPU's predictions"},
            "cpu_baseline": {"ms_per_batch": cpu_t / max(cpu_frames // batch, 1) * 1e3, "frames_per_s": cpu_frames / cpu_t,
                             "edges_per_s": cpu_edges / cpu_t, "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                             "sample": f"{cpu_frames // batch} of the same batches through the oracle chain (numpy graph build, torch CPU "
                                       f"MPN op for op, numpy / scipy post-processing)"}}


def config_leg(nodes, L, edge_state, device, args, steps, cpu_budget_s=6.0, fused_cpu=False):
    """One BASELINE.json config beside the headline: ONE dense `nodes`-node graph, `L` steps,
    3 classified steps, `edge_state` storage of the edge latents (arithmetic fp32 either way).  ms per forward = the faster of the
    one-forward-at-a-time forms (eager calls / one HIP graph of the block's K forwards), edges/s, the step kernel's roofline entry from
    HIP events attached to its launches, max |logit - oracle.TorchOracle| on the same inputs, and that oracle's own time on the host."""
    import copy

    from gnn_cca_amd.inference import GraphedForward
    from oracle.mpn_oracle import TorchOracle
    params = graph_net_params(L=L)
    model = build_model(copy.deepcopy(params), nodes).to(device)
    model.edge_state_dtype = edge_state
    data = make_data(nodes, 1, 1, device)
    E, N = data.edge_index.shape[1], data.x.shape[0]
    by_mode = {}
    with torch.no_grad():
        blocks, out = timed_blocks(lambda: model(data), steps, min(args.warmup, 10), None, device, args.backend, min_blocks=5, min_total_s=0.05)
        by_mode["eager"] = blocks[len(blocks) // 2] / steps * 1e3
        if steps * 4 * min(L, 3) * E <= (1 << 30):
            try:
                gf = GraphedForward(model, warmup=0)
                blk = gf.block([data] * steps, adopt_inputs=True)
                blk.replay()
                blocks, _ = timed_blocks(None, steps, 0, None, device, args.backend, min_blocks=5, min_total_s=0.05, run_block=lambda: blk.replay()[-1])
                by_mode["graph_block"] = blocks[len(blocks) // 2] / steps * 1e3
                del blk, gf
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] config leg dense{nodes}: block capture failed ({type(exc).__name__}: {exc})", file=sys.stderr)
                torch.cuda.synchronize()
        kms = {}
        for _ in range(8):
            for _q in range(2):
                model(data)
            _, times = model.forward_profiled(data)
            for kind, ms in times:
                kms.setdefault(kind, []).append(ms)
        got = [o.detach().cpu() for o in model(data)["classified_edges"]]
    ms = min(by_mode.values())
    e_bytes = 12 if edge_state == "bf16" else 24
    per_launch = step_algorithmic_bytes(E, L, 3, e_bytes=e_bytes)
    alg = float(np.mean(per_launch)) if per_launch else 0.0
    step_us = float(np.mean(kms["step"])) * 1e3 if "step" in kms else float("nan")
    achieved = alg / (step_us * 1e-6) / 1e9 if step_us == step_us and step_us > 0 else 0.0
    # parity + CPU baseline: the reference-shaped torch CPU path on the SAME graph
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    orc = TorchOracle(copy.deepcopy(params), "resnet50", sd)
    dc = make_data(nodes, 1, 1, "cpu")
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    ref = orc.forward(dc.x, dc.edge_index, dc.edge_attr)
    times, t_end = [], time.perf_counter() + cpu_budget_s
    while len(times) < 20 and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        orc.forward(dc.x, dc.edge_index, dc.edge_attr)
        times.append(time.perf_counter() - t0)
    cpu_med = float(np.median(times))
    errs = [float((g - torch.as_tensor(r)).abs().max()) for g, r in zip(got, ref)]
    scale = max(float(torch.as_tensor(r).abs().max()) for r in ref)
    res = {"workload": f"1 x dense{nodes} (N={N}, E={E}), feat 2048, L={L}, 3 classified steps, fp32 arithmetic, {edge_state} edge state, eval",
           "ms_per_step": ms, "ms_per_step_by_mode": by_mode, "value": E / (ms * 1e-3), "unit": "edges/s", "steps": steps,
           "edge_state_storage": "bf16" if edge_state == "bf16" else "f32",
           "roofline": {"bound": "latency" if E * 2 * e_bytes < 32e6 else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "avg_launch_us": step_us, "algorithmic_bytes_per_launch": alg,
                        "kernel": ("mpn_step_fast_kernel" if N <= 512 else "mpn_step_pipe_kernel") + ", mean over the L-1 message steps (HIP events per launch)",
                        "note": ("a step reads and writes %.0f MB of edge state: beyond the 32 MB of L2 but inside the 256 MB Infinity Cache -- `frac` is "
                                 "algorithmic bytes / time / HBM peak, a number to compare runs by, not this launch's bound (floor + a dependent chain)"
                                 % (2 * E * e_bytes / 1e6))
                        if 32e6 <= E * 2 * e_bytes < 256e6 else ""},
           "kernels_us": {k: float(np.mean(v)) * 1e3 for k, v in kms.items()},
           "parity": {"max_abs_err": errs, "max_rel_err": max(errs) / max(scale, 1e-30), "max_abs_logit": scale, "tolerance_abs": 1e-4,
                      "ok": bool(max(errs) <= 1e-4), "against": "oracle.TorchOracle (fp32, CPU) on the same graph"},
           "cpu_baseline": {"value": E / cpu_med, "unit": "edges/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                            "sample": f"{len(times)} forwards of the same graph, median {cpu_med * 1e3:.2f} ms; oracle.TorchOracle, fp32"}}
    res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def dynamic_range_leg(device, nodes=64):
    """`dynamic_range` (SURVEY 7.3 / 8d "report relative error too"): the headline architecture with torch's default initialisation left
    UNconditioned (`sum` aggregation grows the activations ~N per step: max |logit| tens to hundreds instead of 0.5), one dense `nodes`-node
    graph; the HIP logits with fp32 and with bf16 edge-state storage against the fp64 oracle, relative to max |logit|, next to the fp32 oracle's
    own distance from fp64 (the reference arithmetic's gap: the yardstick tests/test_gpu_dynamic_range.py bounds the HIP path by)."""
    import copy

    from oracle.mpn_oracle import NumpyOracle
    params = graph_net_params(L=4)
    model = build_model(copy.deepcopy(params), 2).to(device)          # 1 / (2 - 1): the node MLP is left as initialised
    data = make_data(nodes, 1, 1, device)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    x, ei, ea = data.x.cpu().numpy(), data.edge_index.cpu().numpy(), data.edge_attr.cpu().numpy()
    ref64 = NumpyOracle(copy.deepcopy(params), "resnet50", sd, np.float64).forward(x, ei, ea)
    ref32 = NumpyOracle(copy.deepcopy(params), "resnet50", sd, np.float32).forward(x, ei, ea)
    scale = max(float(np.abs(r).max()) for r in ref64)
    out = {"nodes": nodes, "max_abs_logit": scale,
           "oracle_fp32_rel": max(float(np.abs(a.astype(np.float64) - r).max()) for a, r in zip(ref32, ref64)) / scale}
    for state in ("fp32", "bf16"):
        model.edge_state_dtype = state
        with torch.no_grad():
            got = model(data)["classified_edges"]
        torch.cuda.synchronize()
        out[f"hip_{state}_state_rel"] = max(float(np.abs(o.cpu().numpy().astype(np.float64) - r).max()) for o, r in zip(got, ref64)) / scale
    out["ok"] = bool(out["hip_fp32_state_rel"] <= 4 * out["oracle_fp32_rel"] + 1e-7)
    return out


def train_leg(device, args, frames=64, cams=4, per=5, cpu_reps=3):
    """`train_step`: one training iteration of the reference (train.py:454-494: forward, BCE loss over the classified steps, backward,
    SGD step) on a batch of 64 Terrace-shaped frames (4 cameras x 5 detections, cross-camera edges only; config_training.yaml's model:
    no BatchNorm) through the fused engine -- eager and as gnn_cca_amd.training.GraphedTrainStep's whole-iteration HIP graph --, the
    loss and one gradient against oracle.TorchTrainOracle (the reference's ops under torch autograd on the CPU), and that oracle's time."""
    import copy

    from gnn_cca_amd.training import GraphedTrainStep
    from oracle.mpn_oracle import TorchTrainOracle
    params = graph_net_params(cls_bn=False)
    n_g = cams * per
    cam = np.repeat(np.arange(cams), per)
    li, lj = np.nonzero(cam[:, None] != cam[None, :])
    ei = np.concatenate([np.stack([li, lj]) + f * n_g for f in range(frames)], axis=1).astype(np.int64)
    N, E = frames * n_g, ei.shape[1]
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, 2048)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    ea = rng.random((E, 4)).astype(np.float32)
    lab = (rng.random(E) < 0.2).astype(np.float32)
    model = build_model(copy.deepcopy(params), n_g).to(device)
    model.train()
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    d = Data()
    d.x, d.edge_index, d.edge_attr = torch.from_numpy(x).to(device), torch.from_numpy(ei).to(device), torch.from_numpy(ea).to(device)
    labels = torch.from_numpy(lab).to(device)
    crit = torch.nn.BCEWithLogitsLoss()

    def loss_fn(outputs, lbl):
        return sum(crit(t.view(-1), lbl) for t in outputs["classified_edges"])

    # parity first, from the initial weights: loss and d loss / d (edge-model weight) against the autograd oracle
    model.zero_grad(set_to_none=True)
    loss0 = loss_fn(model(d), labels)
    loss0.backward()
    gname = "MPNet.edge_model.edge_mlp.fc_layers.0.weight"
    g_gpu = dict(model.named_parameters())[gname].grad.detach().cpu().clone()
    loss0 = float(loss0)       # (no tensor of this iteration's autograd graph may outlive it: the whole-iteration capture below would find
    model.zero_grad(set_to_none=True)   # AccumulateGrad nodes of another stream alive and synchronise inside the capture)
    torch.cuda.synchronize()
    engine = getattr(model, "_train_path", None)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    orc = TorchTrainOracle(copy.deepcopy(params), "resnet50", sd0)
    r_loss, _, r_grads = orc.loss_and_grads(x, ei, ea, lab)
    g_ref = torch.as_tensor(r_grads[gname])
    t0 = time.perf_counter()
    for _ in range(cpu_reps):
        orc.loss_and_grads(x, ei, ea, lab)
    cpu_ms = (time.perf_counter() - t0) / cpu_reps * 1e3
    gscale = max(float(g_ref.abs().max()), 1e-30)
    parity = {"loss_gpu": loss0, "loss_oracle": float(r_loss), "loss_abs_err": abs(loss0 - float(r_loss)),
              "grad": gname, "grad_max_abs_err": float((g_gpu - g_ref).abs().max()), "grad_max_abs": gscale,
              "against": "oracle.TorchTrainOracle (torch CPU autograd over the reference's ops) from the same initial weights"}
    parity["ok"] = bool(parity["loss_abs_err"] <= 1e-5 * max(1.0, abs(float(r_loss))) and parity["grad_max_abs_err"] <= 1e-4 * max(gscale, 1e-6) + 1e-7)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)

    gstep = GraphedTrainStep(model, opt, loss_fn, warmup=3)    # its first three calls ARE the eager iteration (GraphedTrainStep._eager)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        gstep(d, labels)
    torch.cuda.synchronize()
    t_first = (time.perf_counter() - t0) / 3 * 1e3

    def eager_step():
        gstep._eager(d, labels)

    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        eager_step()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / reps * 1e3
    graph_ms, graph_finite = None, None
    try:
        for _ in range(3):      # the capture (fourth call through the object) and two replays
            gstep(d, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gl = gstep(d, labels)
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / 50 * 1e3
        graph_finite = bool(torch.isfinite(gl).item())
    except Exception as exc:  # noqa: BLE001
        print(f"[bench] train leg: whole-iteration capture failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        torch.cuda.synchronize()
    del t_first
    best = min(v for v in (eager_ms, graph_ms) if v is not None)
    return {"workload": f"{frames} frames of {cams} cameras x {per} detections as one batch (N={N}, E={E}), feat 2048, L=4, 3 classified steps, "
                        f"config_training.yaml's model (no BatchNorm), BCE loss over the classified steps, SGD",
            "engine": engine, "ms_per_iteration": best, "ms_per_iteration_eager": eager_ms, "ms_per_iteration_graphed": graph_ms,
            "graphed_loss_finite": graph_finite, "iterations_per_s": 1e3 / best, "edges_per_s": E * 1e3 / best, "parity": parity,
            "cpu_baseline": {"ms_per_iteration": cpu_ms, "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                             "sample": f"{cpu_reps} forward + loss + backward passes of the same batch through oracle.TorchTrainOracle (no optimizer step)"},
            "gpu_over_cpu": cpu_ms / best}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=256)
    ap.add_argument("--graphs", type=int, default=1, help="independent graphs per GPU per step")
    ap.add_argument("--L", type=int, default=4)
    ap.add_argument("--mode", choices=["auto", "eager", "graph", "graphk", "graphs"], default="auto",
                    help="eager: one C-ABI call per step; graph: the same call captured once in a HIP graph and replayed per step; "
                         "graphk: the K forwards of a timed block captured back to back in one HIP graph, replayed once per block; "
                         "graphs: the K forwards as HIP graphs of four frames round robin on --streams streams (frames in flight); "
                         "auto: time each form after warm-up and keep the fastest (reported in config.mode)")
    ap.add_argument("--edge-state", choices=["fp32", "bf16"], default="fp32",
                    help="storage of the edge latents between steps (bf16: GNNCCA_OPT_EDGE_STATE_BF16; arithmetic stays fp32)")
    ap.add_argument("--enc-products", type=int, choices=[6, 3], default=6,
                    help="split-bf16 products of the first encoder layer on batches of >= 4096 nodes (3: GNNCCA_OPT_ENC_SPLIT3, "
                         "an accuracy/speed option; the default 6 keeps fp32-level accuracy)")
    ap.add_argument("--enc-unsplit", action="store_true",
                    help="GNNCCA_OPT_ENC_UNSPLIT: forwards over >= 4096 nodes never split K in the first encoder layer (a graph's logits "
                         "are then bitwise independent of its batch / shard; off by default)")
    ap.add_argument("--streams", type=int, default=3, help="forwards in flight of the `pipelined` leg (N = 1, --mode auto; 0 / 1: skip)")
    ap.add_argument("--no-scale-probe", action="store_true", help="skip the 64-graph batch probe of the step kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-terrace", action="store_true", help="skip the EPFL-Terrace frame-distribution pipeline leg (N = 1)")
    ap.add_argument("--no-config4", action="store_true", help="skip the sharded 512 x dense128 leg (BASELINE config 4)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of BASELINE configs 2, 3 (bf16 state) and 5 (N = 1)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-iteration leg (N = 1)")
    ap.add_argument("--train-leg-only", action="store_true", help="(internal) run the training-iteration leg alone and print its JSON object")
    ap.add_argument("--config4-graphs", type=int, default=512)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    ap.add_argument("--profile-reps", type=int, default=20)
    ap.add_argument("--min-blocks", type=int, default=10)
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="launcher: seconds before the rank processes are stopped")
    ap.add_argument("--fail-capture-rank", type=int, default=-1,
                    help="(diagnostic) HIP-graph capture 'fails' on this rank: every rank must then time the same remaining forms (eager) and exit 0")
    args = ap.parse_args()

    if args.train_leg_only:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path is the only implementation (no CPU fallback)")
        torch.cuda.set_device(0)
        out = train_leg(torch.device("cuda", 0), args)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)   # does not return
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, rank 0's JSON: communication libraries print banners from C++ ("[Gloo] Rank 0 is connected to ..."),
    # so for the whole run file descriptor 1 points at stderr and the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(or leave WORLD_SIZE unset and let bench.py start the ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only implementation (no CPU fallback)")
    if args.single_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} device(s) are "
                         f"visible (rehearse on one GPU with --single-device --backend gloo)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    params = graph_net_params(L=args.L)
    # ranks start from DIFFERENT random weights (seed = rank); rank 0's are the job's after the broadcast
    model = build_model(params, args.nodes, seed=rank).to(device)
    model.edge_state_dtype = args.edge_state
    model.encoder_products = args.enc_products
    model.encoder_unsplit = args.enc_unsplit
    # shared weights: ONE RCCL broadcast of rank 0's parameters over xGMI (no other collective on the path)
    broadcast_ms, weights_hash = None, None
    if world > 1:
        broadcast_ms, weights_hash = broadcast_and_verify(model, dist, device, args.backend, rank, world)
    data = make_data(args.nodes, args.graphs, 1 + rank, device)
    E = data.edge_index.shape[1]
    N = data.x.shape[0]

    with torch.no_grad():
        from gnn_cca_amd.inference import GraphedForward   # the product API of the graph forms (inference.py:173-283's per-frame loop)

        def eager_run():
            return model(data)

        # The three issue forms of a block of K steps, each through the package API a caller would use:
        #   eager        K calls of MOTMPNet.forward (one C-ABI call, six launches, per step)
        #   graph        K replays of GraphedForward's one-forward HIP graph; the frame lives in the graph's static input buffers
        #                (`static_inputs`: the producer writes each frame there), so a replay copies nothing
        #   graph_block  ONE replay of GraphedForward.block([frame] * K, adopt_inputs=True): the K forwards -- every launch of each,
        #                each with its own outputs -- captured back to back in one HIP graph
        # `auto` times all three with the same block protocol, reports each in config.ms_per_step_by_mode and keeps the fastest of
        # these ONE-FORWARD-AT-A-TIME forms as `value` (edges/s = E / t_forward -- the definition of rounds 1-3).  The
        # frames-in-flight form (graph_block_chains, below) is timed too and reported under `pipelined`; it is never `value`.
        forms = {"eager": (eager_run, None)}
        gf = GraphedForward(model, warmup=0)
        if args.mode in ("graph", "graphk", "graphs", "auto"):
            try:
                if rank == args.fail_capture_rank:
                    raise RuntimeError("--fail-capture-rank: simulated capture failure on this rank")
                static = gf.static_inputs(data)
                static.x.copy_(data.x), static.edge_index.copy_(data.edge_index), static.edge_attr.copy_(data.edge_attr)
                gf(static)
                forms["graph"] = (lambda: gf(static), None)
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); using eager launches", file=sys.stderr)
                torch.cuda.synchronize()
            out_bytes = 4 * min(args.L, 3) * E
            if "graph" in forms and args.mode in ("graphk", "graphs", "auto") and args.steps * out_bytes <= (1 << 30):
                try:
                    blk = gf.block([data] * args.steps, adopt_inputs=True)
                    blk.replay()
                    forms["graph_block"] = (None, lambda: blk.replay()[-1])   # (the block stays cached in gf)
                except Exception as exc:  # noqa: BLE001
                    print(f"[bench] HIP graph capture of a {args.steps}-step block failed ({type(exc).__name__}: {exc})", file=sys.stderr)
                    torch.cuda.synchronize()
            # graph_block_chains: the same K forwards cut into HIP graphs of four frames, replayed round robin on S streams
            # (GraphedForward.block(..., chains=S, depth=4)): S independent frames in flight, a workspace per stream, every forward
            # complete with its own outputs -- the throughput form of the per-frame loop.  (Measured on the way: S parallel branches
            # inside ONE graph 25.8 us, one big graph per stream 26.1 us -- the host-side launch of a 400-kernel graph takes a third
            # of its run --, one-forward graphs round robin 15.9-26 us depending on how the streams land on the hardware queues.)
            if "graph_block" in forms and args.streams > 1 and args.mode in ("graphs", "auto"):
                try:
                    # frames per graph: four on long blocks; two / one on short ones so that every stream gets the same number of groups
                    depth_s = 4 if args.steps >= 16 * args.streams else (2 if args.steps >= 4 * args.streams else 1)
                    blk_s = gf.block([data] * args.steps, adopt_inputs=True, chains=args.streams, depth=depth_s)
                    blk_s.replay()
                    forms["graph_block_chains"] = (None, lambda: blk_s.replay()[-1])
                except Exception as exc:  # noqa: BLE001
                    print(f"[bench] capture of the {args.streams}-stream block failed ({type(exc).__name__}: {exc})", file=sys.stderr)
                    torch.cuda.synchronize()
        want = {"eager": ["eager"], "graph": ["graph"], "graphk": ["graph_block"], "graphs": ["graph_block_chains"],
                "auto": ["eager", "graph", "graph_block", "graph_block_chains"]}[args.mode]
        # every rank times the SAME forms, in the same order (a capture that failed on one rank drops the form on all of them)
        agreed = agree_forms(forms, dist, device, args.backend)
        if rank == 0 and sorted(agreed) != sorted(forms):
            print(f"[bench] forms dropped because a rank could not capture them: {sorted(set(forms) - set(agreed))}", file=sys.stderr)
        timed = {}
        for name in want:
            if name not in agreed:
                continue
            run_f, run_b = forms[name]
            timed[name] = timed_blocks(run_f, args.steps, args.warmup, dist, device, args.backend, min_blocks=args.min_blocks,
                                       run_block=run_b)
        if not timed:   # the requested graph form could not be captured
            timed["eager"] = timed_blocks(eager_run, args.steps, args.warmup, dist, device, args.backend, min_blocks=args.min_blocks)
        by_mode = {k: v[0][len(v[0]) // 2] / args.steps * 1e3 for k, v in timed.items()}
        # `value` = one forward at a time; frames in flight only when that form was asked for by name (--mode graphs)
        one_at_a_time = {k: v for k, v in by_mode.items() if k != "graph_block_chains"} or by_mode
        best = min(one_at_a_time, key=one_at_a_time.get)
        blocks, out = timed[best]
        api = {"eager": "eager: MOTMPNet.forward, one call per step",
               "graph": "graph: GraphedForward.__call__, one HIP-graph replay per step",
               "graph_block": f"graph_block: GraphedForward.block, one HIP graph of {args.steps} forwards",
               "graph_block_chains": f"graph_block_chains: GraphedForward.block(chains={args.streams}), {args.streams} frames in flight"}
        mode_used = api[best] + (" (auto: fastest one-at-a-time form)" if args.mode == "auto" else "")
        run = forms[best][0] or eager_run
        t = blocks[len(blocks) // 2]      # median block of K steps (max over ranks inside every block)
        # one frame at a time vs frames in flight, side by side (both are in ms_per_step_by_mode; `value` is the one-at-a-time form)
        pipelined = None
        if "graph_block_chains" in by_mode and "graph_block" in by_mode:
            same = all(torch.equal(a_, b_) for a_, b_ in zip(timed["graph_block_chains"][1]["classified_edges"],
                                                             timed["graph_block"][1]["classified_edges"]))
            pipelined = {"frames_in_flight": args.streams, "ms_per_forward_in_flight": by_mode["graph_block_chains"],
                         "ms_per_forward_one_at_a_time": by_mode["graph_block"], "bitwise_equal_outputs": bool(same),
                         "note": "one at a time = the latency of a forward inside a HIP-graph block; in flight = whole-job throughput of "
                                 "independent frames (every forward complete, its own outputs, a workspace per stream)"}
        # every rank's OWN time per step (no barrier inside), gathered: how evenly the ranks run
        rank_ms = None
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run()
            torch.cuda.synchronize()
            own = (time.perf_counter() - t0) / args.steps * 1e3
            dev_ = device if args.backend == "nccl" else "cpu"
            all_ms = [torch.zeros(1, dtype=torch.float64, device=dev_) for _ in range(world)]
            dist.all_gather(all_ms, torch.tensor([own], dtype=torch.float64, device=dev_))
            rank_ms = [float(v.item()) for v in all_ms]
        ok = all(torch.isfinite(o).all().item() for o in out["classified_edges"])

        # ---- BASELINE config 4: 512 independent dense128 graphs, sharded 512/N per rank (forward_sharded) ----------
        cfg4, share = None, None
        # (the leg's timed blocks are collective: every rank must enter it or none -- eligibility is a pure function of the arguments, and
        # is all-reduced anyway so that a rank with a different command line cannot desynchronise the job)
        if agree_flag(not args.no_config4 and args.config4_graphs >= 1, dist, device, args.backend):
            from gnn_cca_amd.sharding import forward_sharded, shard_batch
            m4 = build_model(graph_net_params(L=4), 128, seed=rank).to(device)
            m4.edge_state_dtype = args.edge_state
            m4.encoder_products = args.enc_products
            m4.encoder_unsplit = args.enc_unsplit
            if world > 1:
                broadcast_and_verify(m4, dist, device, args.backend, rank, world)
            graphs4 = LazyDenseGraphs(args.config4_graphs, 128, device)
            lo4, hi4, batch4 = shard_batch(graphs4, rank, world)       # this rank's union, resident in HBM
            e4_local = batch4.edge_index.shape[1] if batch4 is not None else 0
            e4_total = args.config4_graphs * 128 * 127
            steps4 = max(1, min(args.steps, 50))
            blocks4, res4 = timed_blocks(lambda: forward_sharded(m4, graphs4, rank, world, batch=batch4), steps4,
                                         min(args.warmup, 10), dist, device, args.backend, min_blocks=5, min_total_s=0.1)
            t4_eager = blocks4[len(blocks4) // 2]
            # the same forward as ONE HIP-graph replay per step (forward_sharded(..., graphed=GraphedForward): captured on the resident union's
            # own tensors, no copies; bit for bit the eager logits).  Timed on every rank or on none (the capture's success is all-reduced).
            gf4, t4_graph = None, None
            try:
                if rank == args.fail_capture_rank:
                    raise RuntimeError("--fail-capture-rank: simulated capture failure on this rank")
                gf4 = GraphedForward(m4, warmup=0)
                if batch4 is not None:
                    forward_sharded(m4, graphs4, rank, world, batch=batch4, graphed=gf4)
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] config 4: HIP graph capture failed on rank {rank} ({type(exc).__name__}: {exc})", file=sys.stderr)
                gf4 = None
                torch.cuda.synchronize()
            if agree_flag(gf4 is not None, dist, device, args.backend):
                blocks4g, res4g = timed_blocks(lambda: forward_sharded(m4, graphs4, rank, world, batch=batch4, graphed=gf4), steps4,
                                               min(args.warmup, 10), dist, device, args.backend, min_blocks=5, min_total_s=0.1)
                t4_graph = blocks4g[len(blocks4g) // 2]
                same4 = all(torch.equal(a_, b_) for ga, gb in zip(res4[2][:2], res4g[2][:2]) for a_, b_ in zip(ga, gb))
                if t4_graph < t4_eager:
                    blocks4, res4 = blocks4g, res4g
            t4 = blocks4[len(blocks4) // 2]
            ok4 = all(torch.isfinite(o).all().item() for g in res4[2][:2] for o in g)
            k4 = {}
            if batch4 is not None:   # this rank's share, kernel by kernel (events attached to every dispatch)
                for _ in range(5):
                    _, times4 = m4.forward_profiled(batch4)
                    for kind, ms in times4:
                        k4.setdefault(kind, []).append(ms)
            cfg4 = {"workload": f"{args.config4_graphs} x dense128 graphs (E={e4_total}), sharded {hi4 - lo4} per rank "
                                f"through sharding.forward_sharded, L=4, 3 classified steps, fp32, eval",
                    "value": e4_total * steps4 / t4, "unit": "edges/s", "scaling": "strong", "n_gpus": world,
                    "graphs_per_rank": hi4 - lo4, "edges_rank0": e4_local, "steps": steps4, "blocks": len(blocks4),
                    "ms_per_step": t4 / steps4 * 1e3, "outputs_finite": bool(ok4),
                    "ms_per_step_eager": t4_eager / steps4 * 1e3, "ms_per_step_graph": (t4_graph / steps4 * 1e3) if t4_graph else None,
                    "graph_equals_eager_bitwise": bool(same4) if t4_graph else None,
                    "mode": "graph: one HIP-graph replay per forward" if t4_graph and t4_graph < t4_eager else "eager",
                    "kernels_us_rank0": {k: float(np.mean(v)) * 1e3 for k, v in k4.items()}}
            if "step" in k4 and e4_local > 0:
                # this rank's message steps against HBM: the step kernel at its config-4 operating point (batches of >= 16 384 nodes of
                # degree <= 128 run two nodes per wave with the classification deferred -- DESIGN.md section 5 -- so the message steps
                # write ONE of the two classified states' logits; the last step, not in this mean, writes the other two)
                n4 = (hi4 - lo4) * 128
                deferred = n4 >= 16384
                per = [b for b in step_algorithmic_bytes(e4_local, 4, 3)]
                if deferred:
                    per[1] -= 4 * e4_local
                alg4 = float(np.mean(per))
                us4 = float(np.mean(k4["step"])) * 1e3
                cfg4["roofline_rank0"] = {"bound": "hbm", "kernel": "mpn_step_pipe_kernel, mean over the three message steps of this rank's forward",
                                          "avg_launch_us": us4, "algorithmic_bytes_per_launch": alg4, "achieved": alg4 / (us4 * 1e-6) / 1e9,
                                          "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg4 / (us4 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                          "two_nodes_per_wave_and_deferred_classification": bool(deferred)}
                fr4 = leg_fractions(cfg4["kernels_us_rank0"], n4, e4_local, cfg4["ms_per_step"], alg4)
                cfg4["enc_frac_rank0"], cfg4["forward_frac_rank0"] = fr4.get("enc_frac"), fr4.get("forward_frac")
            # N = 1: the per-GPU share of the 8-GPU run of the same job (graphs [0, 512/8) of the SAME lazy sequence, as rank 0 of 8
            # would build it) timed on this GPU, same box, same run: the only strong-scaling evidence obtainable without an 8-GPU
            # node -- projected_8gpu_speedup = union ms / share ms (no collective on the data path; the broadcast is one-time)
            share = None
            if world == 1 and args.config4_graphs % 8 == 0:
                lo8, hi8, batch8 = shard_batch(graphs4, 0, 8)
                blocks8, _ = timed_blocks(lambda: forward_sharded(m4, graphs4, 0, 8, batch=batch8), steps4, min(args.warmup, 10), None,
                                          device, args.backend, min_blocks=5, min_total_s=0.1)
                t8_eager = t8 = blocks8[len(blocks8) // 2]
                t8_graph = None
                if gf4 is not None:       # the share the same way as the union: one HIP-graph replay per forward
                    try:
                        forward_sharded(m4, graphs4, 0, 8, batch=batch8, graphed=gf4)
                        blocks8g, _ = timed_blocks(lambda: forward_sharded(m4, graphs4, 0, 8, batch=batch8, graphed=gf4), steps4, min(args.warmup, 10),
                                                   None, device, args.backend, min_blocks=5, min_total_s=0.1)
                        t8_graph = blocks8g[len(blocks8g) // 2]
                        t8 = min(t8, t8_graph)
                    except Exception as exc:  # noqa: BLE001
                        print(f"[bench] config 4 share: HIP graph capture failed ({type(exc).__name__}: {exc})", file=sys.stderr)
                        torch.cuda.synchronize()
                k8 = {}
                for _ in range(5):
                    _, times8 = m4.forward_profiled(batch8)
                    for kind, ms in times8:
                        k8.setdefault(kind, []).append(ms)
                share = {"workload": f"{hi8 - lo8} x dense128 graphs (E={batch8.edge_index.shape[1]}): rank 0's share of "
                                     f"{args.config4_graphs} graphs over 8 GPUs, run on this one GPU",
                         "ms_per_step": t8 / steps4 * 1e3, "union_ms_per_step": t4 / steps4 * 1e3,
                         "ms_per_step_eager": t8_eager / steps4 * 1e3, "ms_per_step_graph": (t8_graph / steps4 * 1e3) if t8_graph else None,
                         "projected_8gpu_speedup": t4 / t8, "projected_8gpu_speedup_eager": t4_eager / t8_eager, "target": 6.0,
                         "kernels_us": {k: float(np.mean(v)) * 1e3 for k, v in k8.items()},
                         "note": "projection from one-GPU measurements, not an 8-GPU measurement: every rank runs this share "
                                 "concurrently with no data-path collective"}
                e8 = batch8.edge_index.shape[1]
                share.update(leg_fractions(share["kernels_us"], (hi8 - lo8) * 128, e8, share["ms_per_step"],
                                           float(np.mean(step_algorithmic_bytes(e8, 4, 3)))))
                del batch8
            del m4, graphs4, batch4, res4, gf4

        # per-kernel durations (HIP events attached to every dispatch; separate pass so the timed region is undisturbed)
        kernel_ms = {}
        for _ in range(args.profile_reps):
            for _q in range(4):  # a warm queue, as in the timed region
                model(data)
            _, times = model.forward_profiled(data)
            for kind, ms in times:
                kernel_ms.setdefault(kind, []).append(ms)

    if rank == 0:
        e_bytes = 12 if args.edge_state == "bf16" else 24
        per_launch = step_algorithmic_bytes(E, args.L, 3, e_bytes=e_bytes)
        step_ms = float(np.mean(kernel_ms["step"])) if "step" in kernel_ms else float("nan")
        alg = float(np.mean(per_launch)) if per_launch else 0.0
        achieved = alg / (step_ms * 1e-3) / 1e9 if step_ms == step_ms and step_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the PMC passes (tools/collect_profiles.py -> profiles/):
        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 correction applied
        traffic, rocprof_us = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_index.json")) as f:
                ent = json.load(f).get(f"{args.graphs}x{args.nodes}_L{args.L}" + ("_bf16" if args.edge_state == "bf16" else ""))
            if ent:   # like for like: the call-weighted mean over the message variants, the mix `algorithmic_bytes_per_launch` averages
                traffic = ent.get("hbm_bytes_per_launch_msg_mean", ent["hbm_bytes_per_launch"])
                rocprof_us = ent.get("rocprof_avg_us_msg_mean", ent["rocprof_avg_us"])
        except OSError:
            pass
        # the edge state of one step (read + write) against the 8 x 4 MB of L2: below that the launch is bounded by its
        # dependent round trips after the kernel boundary, not by HBM; the HBM fraction is still reported
        cache_resident = E * 2 * e_bytes < 32e6
        shape = f"{args.nodes}-node dense graph" if args.graphs == 1 else f"{args.graphs} x {args.nodes}-node dense graphs"
        res = {
            "metric": f"processed edges/sec (L={args.L} MPN steps), {shape}",
            "value": world * E * args.steps / t,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",   # the arithmetic type of the path (every product and sum is fp32-accurate; see config.edge_state for storage)
            "data": "synthetic",
            "config": {"workload": f"{args.graphs} x dense{args.nodes} per GPU per step (N={N}, E={E}), feat 2048, L={args.L}, 3 cls steps, "
                                   f"fp32, {args.edge_state} state, eval",
                       "mode": mode_used, "ms_per_step_by_mode": by_mode,
                       "frames_in_flight": args.streams if best == "graph_block_chains" else 1, "edge_state": args.edge_state, "edge_state_storage": "bf16" if args.edge_state == "bf16" else "f32",
                       "encoder_products": args.enc_products, "encoder_unsplit": args.enc_unsplit,
                       "outputs_finite": bool(ok),
                       "edge_steps_per_s": world * E * args.L * args.steps / t,
                       "timing": f"median of {len(blocks)} blocks of {args.steps} steps (barrier + sync both sides, MAX over ranks)",
                       "block_ms": {"min": blocks[0] * 1e3, "median": t * 1e3, "max": blocks[-1] * 1e3},
                       "launcher": "single process" if world == 1 else "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                       else "bench.py (self-launched rank processes)", "backend": args.backend if world > 1 else None,
                       "ranks_seen": dist.get_world_size() if dist else 1, "broadcast_ms": broadcast_ms,
                       "weights_hash": f"{weights_hash & 0xFFFFFFFFFFFFFFFF:016x}" if weights_hash is not None else None,
                       "rank_ms_per_step": ({"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms} if rank_ms else None)},
            "roofline": {"bound": "latency" if cache_resident else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("mpn_step_fast_kernel" if N <= 512 else "mpn_step_pipe_kernel") + "<FIRST|CLS, MSG>: message step, L-1 launches/forward",
                         "avg_launch_us": step_ms * 1e3, "rocprof_avg_launch_us": rocprof_us,
                         "algorithmic_bytes_per_launch": alg,
                         "note": "edge state is L2/Infinity-Cache resident at this size: the launch is bounded by its dependent "
                                 "round trips (latency), `frac` is still algorithmic bytes / time / HBM peak; "
                                 "`roofline_at_scale` is the same kernel where HBM is the bound" if cache_resident else ""},
            "kernels_us": {k: float(np.mean(v)) * 1e3 for k, v in kernel_ms.items()},
            "kernels_us_note": "per-kernel times of a SEPARATE eager pass with start / stop events attached to each dispatch "
                               "(gnncca_mpn_forward_profiled); a dispatch's span includes its own launch latency, which consecutive "
                               "kernels of a HIP-graph block overlap -- with config.mode = 'graph of K forwards per block' ms_per_step "
                               "can therefore be below the sum of these",
        }
        if pipelined is not None:
            res["pipelined"] = pipelined
        if cfg4 is not None:
            res["config4_sharded"] = cfg4
            if share is not None:
                res["config4_share"] = share
        if world == 1 and args.graphs == 1 and not args.no_scale_probe:
            res["roofline_at_scale"] = scale_probe(params, device, args)
        if world == 1 and not args.no_terrace:
            try:
                res["terrace_pipeline"] = terrace_leg(device, args)
            except Exception as exc:  # noqa: BLE001
                res["terrace_pipeline"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_configs:
            # BASELINE.json configs 2, 3 as named (bf16 storage) and 5 (fp32 and bf16 storage), each one forward at a time
            res["configs"] = {}
            for key, (nn_, l_, es_) in {"config2_dense64_L4_fp32": (64, 4, "fp32"), "config3_dense256_L4_bf16_state": (256, 4, "bf16"),
                                        "config5_dense1024_L8_fp32": (1024, 8, "fp32"), "config5_dense1024_L8_bf16_state": (1024, 8, "bf16")}.items():
                try:
                    res["configs"][key] = config_leg(nn_, l_, es_, device, args, steps=max(1, min(args.steps, 50 if nn_ >= 1024 else 200)))
                except Exception as exc:  # noqa: BLE001
                    res["configs"][key] = {"error": f"{type(exc).__name__}: {exc}"}
                    torch.cuda.synchronize()
        if world == 1 and not args.no_configs:
            try:
                res["dynamic_range"] = dynamic_range_leg(device)
            except Exception as exc:  # noqa: BLE001
                res["dynamic_range"] = {"error": f"{type(exc).__name__}: {exc}"}
                torch.cuda.synchronize()
        if world == 1 and not args.no_train:
            # in a child process of its own: the leg captures torch's autograd + optimizer into a HIP graph, and a crash inside that
            # machinery must cost this leg, not the line (the parent waits; one process on the GPU at a time computes)
            import subprocess
            try:
                torch.cuda.synchronize()
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--train-leg-only"], capture_output=True, text=True, timeout=600)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                res["train_step"] = json.loads(line[-1]) if cp.returncode == 0 and line else {
                    "error": f"child exit code {cp.returncode}: {cp.stderr.strip().splitlines()[-1] if cp.stderr.strip() else 'no output'}"}
            except Exception as exc:  # noqa: BLE001
                res["train_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(params, model, args.nodes, args.graphs)
            res["parity"] = res["cpu_baseline"].pop("parity", None)
            res["config"]["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            try:   # informational second flavour; the >= 50x target is stated against `cpu_baseline` (the reference-shaped path)
                res["cpu_baseline_fused"] = cpu_baseline_fused(params, model, args.nodes, args.graphs)
            except Exception as exc:  # noqa: BLE001
                res["cpu_baseline_fused"] = {"error": f"{type(exc).__name__}: {exc}"}
        res["config"]["forms_timed"] = ",".join(k for k in FORM_ORDER if k in by_mode)
        flat_evidence(res)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

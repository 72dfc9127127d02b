#!/usr/bin/env python3
"""Runs on the GPU box: profiles `python3 bench.py <args>` with rocprofv3 in separate passes (--kernel-trace --stats; --pmc
FETCH_SIZE; --pmc WRITE_SIZE; two --pmc passes of SQ / GRBM counters -- PMC never combined with tracing) and writes a
summary under gpurun_out/<tag>/: kernel_stats.csv (our kernels only), pmc_summary.json with, per kernel and launch,
  * HBM traffic, corrected as MI355X_MICROARCH.md prescribes and as tools/ubench_calib.hip confirms on this project's access
    pattern: FETCH_SIZE counts exactly half of the bytes read, WRITE_SIZE is exact; both in KiB;
  * matrix-pipe and VALU utilisation: kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs);
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)  (the counter counts cycles, summed over SIMDs);
    valu_util = 4 x SQ_ACTIVE_INST_VALU / (kernel cycles x 1024)          (SQ_ACTIVE_INST_* count quad-cycles);
    shares of the waves' lifetime: issuing (SQ_ACTIVE_INST_ANY), issue-stalled (SQ_WAIT_INST_ANY), parked on s_waitcnt /
    barriers (SQ_WAIT_ANY), each / SQ_WAVE_CYCLES.

    python3 tools/collect_profiles.py <tag> -- <bench.py args...>
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tag, sub, extra, bench_args):
    out = os.path.join(ROOT, "gpurun_out", tag, sub)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3"] + extra + ["--output-format", "csv", "-d", out, "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
    with open(os.path.join(out, "run.log"), "w") as log:
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=False, timeout=240)
        except subprocess.TimeoutExpired:
            print(f"pass {sub} timed out", flush=True)
    return out


def main():
    tag = sys.argv[1]
    trace_only = "--trace-only" in sys.argv[2:sys.argv.index("--")]   # kernel-trace + stats only (e.g. the HIP-graph block form of the headline)
    bench_args = sys.argv[sys.argv.index("--") + 1:] + ["--no-cpu-baseline", "--no-scale-probe", "--no-config4", "--no-terrace", "--no-configs", "--no-train", "--streams", "0", "--profile-reps", "0"]
    summary = {"command": "python3 bench.py " + " ".join(bench_args), "kernels": {}}
    d = run(tag, "trace", ["--kernel-trace", "--stats"], bench_args)
    stats = glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if "gnncca" in r["Name"]]
        with open(os.path.join(ROOT, "gpurun_out", tag, "kernel_stats.csv"), "w") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
        for r in keep:
            summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
    groups = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
               "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"],
              ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
               "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"]]
    for gi, grp in enumerate([] if trace_only else groups):
        d = run(tag, "pmc%d" % gi, ["--pmc"] + grp, bench_args)
        acc = {}
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "gnncca" in r["Kernel_Name"]:
                    acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, ctr), v in acc.items():
            e = summary["kernels"].setdefault(k, {})
            e[ctr + ("_KiB_mean" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else "")] = sum(v) / len(v)
    for k, e in summary["kernels"].items():
        if "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024.0
        if e.get("GRBM_GUI_ACTIVE", 0) > 0:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            e["kernel_cycles"] = cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            if "SQ_ACTIVE_INST_VALU" in e:
                e["valu_util"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0)
        if e.get("SQ_WAVE_CYCLES", 0) > 0:
            for name, ctr in (("wave_share_issuing", "SQ_ACTIVE_INST_ANY"), ("wave_share_issue_stalled", "SQ_WAIT_INST_ANY"),
                              ("wave_share_waiting", "SQ_WAIT_ANY")):
                if ctr in e:
                    e[name] = e[ctr] / e["SQ_WAVE_CYCLES"]
    with open(os.path.join(ROOT, "gpurun_out", tag, "pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()

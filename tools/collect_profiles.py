#!/usr/bin/env python3
"""Runs on the GPU box: profiles `python3 bench.py <args>` with rocprofv3 in three separate passes
(--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- PMC never combined with tracing) and writes a
summary under gpurun_out/<tag>/: kernel_stats.csv (our kernels only), pmc_summary.json (per-kernel mean HBM
traffic per launch, corrected as MI355X_MICROARCH.md prescribes and as tools/ubench_calib.hip confirms on this
project's access pattern: FETCH_SIZE counts exactly half of the bytes read, WRITE_SIZE is exact; both in KiB).

    python3 tools/collect_profiles.py <tag> -- <bench.py args...>
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tag, sub, extra, bench_args):
    out = os.path.join(ROOT, "gpurun_out", tag, sub)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3"] + extra + ["--output-format", "csv", "-d", out, "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
    with open(os.path.join(out, "run.log"), "w") as log:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=False)
    return out


def main():
    tag = sys.argv[1]
    bench_args = sys.argv[sys.argv.index("--") + 1:] + ["--no-cpu-baseline", "--no-scale-probe", "--no-config4", "--profile-reps", "0"]
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
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = run(tag, ctr, ["--pmc", ctr], bench_args)
        files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        acc = {}
        if files:
            for r in csv.DictReader(open(files[0])):
                if "gnncca" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            e = summary["kernels"].setdefault(k, {})
            e[ctr + "_KiB_mean"] = sum(v) / len(v)
    for k, e in summary["kernels"].items():
        if "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024.0
    with open(os.path.join(ROOT, "gpurun_out", tag, "pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Runs on the GPU box: hardware counters of ONE kernel of `python3 bench.py <args>`, one rocprofv3 --pmc pass per
counter group (PMC passes are never combined with tracing), mean per launch.

    python3 tools/pmc_kernel.py <kernel-substring> "CTR_A,CTR_B;CTR_C,CTR_D" -- <bench.py args...>
"""
import csv
import glob
import json
import os
import shutil
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


MAX_PER_BLOCK = 4   # TCP_* / TA_* counters one pass can hold on gfx950: a pass that asks for more (eight were tried in round 4:
                    # profiles/r04_logs/pmc_r04_tcp.log) makes rocprofv3 abort inside the profiled GPU process and hang in finalisation


def split_groups(groups):
    """A group with more than MAX_PER_BLOCK counters of one of those blocks is never handed to rocprofv3: it is cut into passes that fit."""
    out = []
    for grp in groups:
        cur, per_block = [], {}
        for c in grp:
            block = c.split("_", 1)[0] if c.startswith(("TCP_", "TA_")) else None
            if block and per_block.get(block, 0) >= MAX_PER_BLOCK:
                out.append(cur)
                cur, per_block = [], {}
            cur.append(c)
            if block:
                per_block[block] = per_block.get(block, 0) + 1
        if cur:
            out.append(cur)
    if out != groups:
        print(f"pmc_kernel: counter groups re-cut to at most {MAX_PER_BLOCK} TCP_* / TA_* counters per pass: {out}", flush=True)
    return out


def main():
    kern, groups = sys.argv[1], split_groups([g.split(",") for g in sys.argv[2].split(";")])
    bench_args = sys.argv[sys.argv.index("--") + 1:] + ["--no-cpu-baseline", "--no-scale-probe", "--no-config4", "--no-terrace", "--no-configs", "--no-train", "--streams", "0", "--profile-reps", "0", "--mode", "eager",
                                                         "--steps", "30", "--warmup", "5"]
    res = {}
    for gi, grp in enumerate(groups):
        out = os.path.join(ROOT, "gpurun_out", "pmc_kernel", f"g{gi}")
        shutil.rmtree(out, ignore_errors=True)   # a previous run's CSVs must not be averaged in
        os.makedirs(out, exist_ok=True)
        cmd = ["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", out, "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
        with open(os.path.join(out, "run.log"), "w") as log:
            # own process group + hard limit: a counter set the hardware refuses makes rocprofv3 abort and then hang
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT,
                                    start_new_session=True)
            try:
                proc.wait(timeout=int(os.environ.get('PMC_PASS_TIMEOUT', '150')))
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                print(f"group {gi} killed at its limit: {grp}; last lines of rocprofv3's output:", flush=True)
                print("".join(open(os.path.join(out, "run.log"), errors="replace").readlines()[-12:]), flush=True)
                continue
        for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
            acc = {}
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                res[k] = sum(v) / len(v)
        print(f"group {gi} done: {grp}", flush=True)
    print(json.dumps(res, indent=1))
    with open(os.path.join(ROOT, "gpurun_out", "pmc_kernel", "summary.json"), "w") as f:
        json.dump({"kernel": kern, "bench_args": bench_args, "mean_per_launch": res}, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/r<NN>_*_pmc_summary.json (the newest round that has the workload) -> profiles/pmc_index.json, the table bench.py reads `roofline.traffic` and
`rocprof_avg_launch_us` from (keyed "<graphs>x<nodes>_L<L>[_bf16]").  The dominant kernel is the middle message-passing
step, mpn_step_{pipe,persist}_kernel<FIRST=false, CLS=true, MSG=true, ...> (rounds 1-2: mpn_step_fast_kernel).  Next to it the
call-weighted MEAN over all message variants (<*, *, MSG=true>): the mix `roofline.algorithmic_bytes_per_launch` averages."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
index = {}
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_pmc_summary.json"))):  # later rounds overwrite
    s = json.load(open(path))
    cmd = s["command"]
    def arg(name, default):
        m = re.search(rf"--{name} (\S+)", cmd)
        return m.group(1) if m else default
    key = f"{arg('graphs', '1')}x{arg('nodes', '256')}_L{arg('L', '4')}" + ("_bf16" if arg("edge-state", "f32") == "bf16" else "")
    best, mix_b, mix_t, mix_c = None, 0.0, 0.0, 0
    for name, e in s["kernels"].items():
        if not re.search(r"mpn_step_(fast|pipe|persist)_kernel<", name) or "hbm_bytes_per_launch" not in e:
            continue
        if re.search(r"kernel<false, true, true", name) and (best is None or e["calls"] > best[1]["calls"]):
            best = (name, e)
        if re.search(r"kernel<(true|false), (true|false), true", name):   # every variant with the message block
            mix_b += e["hbm_bytes_per_launch"] * e["calls"]
            mix_t += e["avg_ns"] * e["calls"]
            mix_c += e["calls"]
    if best is None:   # deferred classification (round 4): the classifying message step is the variant that classifies its INPUT, <false, false, true, ..., true>
        for name, e in s["kernels"].items():
            if re.search(r"mpn_step_pipe_kernel<false, false, true(, \w+)*, true>", name) and "hbm_bytes_per_launch" in e:
                best = (name, e)
    if best:
        index[key] = {"kernel": best[0], "hbm_bytes_per_launch": best[1]["hbm_bytes_per_launch"],
                      "rocprof_avg_us": best[1]["avg_ns"] / 1e3, "source": os.path.relpath(path, ROOT)}
        if mix_c:
            index[key]["hbm_bytes_per_launch_msg_mean"] = mix_b / mix_c
            index[key]["rocprof_avg_us_msg_mean"] = mix_t / mix_c / 1e3
        for extra in ("mfma_util", "valu_util", "wave_share_issuing", "wave_share_issue_stalled", "wave_share_waiting"):
            if extra in best[1]:
                index[key][extra] = best[1][extra]
json.dump(index, open(os.path.join(ROOT, "profiles", "pmc_index.json"), "w"), indent=1)
print(json.dumps(index, indent=1))

#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in libgnncca_mpn (compile-only, runs without a GPU).
usage: python tools/kernel_regs.py [substring ...]     e.g.  python tools/kernel_regs.py mpn_step_pipe_kernel"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/gnncca_regs.txt"
if "--reuse" not in sys.argv or not os.path.exists(out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-I", os.path.join(ROOT, "include"), "-I",
           os.path.join(ROOT, "gnn-cca_amd", "csrc"), "-DGNNCCA_BUILD", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-c",
           os.path.join(ROOT, "gnn-cca_amd", "csrc", "mpn_forward.hip"), "-o", "/tmp/gnncca_dev.o", "-Rpass-analysis=kernel-resource-usage"]
    with open(out, "w") as f:
        subprocess.run(cmd, stderr=f, check=True)
want = [a for a in sys.argv[1:] if not a.startswith("--")]
rows, cur = {}, None
for line in open(out):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    short = d.replace("void gnncca::", "").split("(")[0]
    if want and not any(w in short for w in want):
        continue
    r = rows[n]
    print(f"{short:95s} VGPR {r.get('VGPRs'):4d} SGPR {r.get('TotalSGPRs'):4d} scratch {r.get('ScratchSize'):4d} spillV {r.get('VGPRs Spill'):3d} "
          f"spillS {r.get('SGPRs Spill'):3d} occ {r.get('Occupancy')}")

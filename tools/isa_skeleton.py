#!/usr/bin/env python3
"""The memory / wait skeleton of one kernel's ISA: every s_load, vector load, LDS store, s_waitcnt, s_barrier and branch in program order, with line
numbers -- how round 5 found the step kernels' prologue problems (a staging load sunk into a branch behind `s_waitcnt vmcnt(0)`, a `vmcnt(0)` in
front of the column-range arm, the BAD_INDEX test hoisted above the CSR loads, kernel arguments fetched cluster by cluster).  No GPU needed.
    python tools/isa_skeleton.py 'mpn_step_pipe_kernel<false, true, true, false, false, 0, false, 1, false>' [first_lines=120] [--all]
Compiles csrc/mpn_forward.hip to /tmp/gnncca_mpn_forward.s once (about 80 s; --reuse keeps an existing listing), prints the kernel's register /
scratch figures from the code-object metadata, then the skeleton of its first `first_lines` instructions (--all: every instruction kind)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LISTING = "/tmp/gnncca_mpn_forward.s"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = args[0]
first = int(args[1]) if len(args) > 1 else 120
if "--reuse" not in sys.argv or not os.path.exists(LISTING):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "gnn-cca_amd", "csrc"), "-DGNNCCA_BUILD", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "gnn-cca_amd", "csrc", "mpn_forward.hip"), "-o", LISTING], check=True, stderr=subprocess.DEVNULL)
text = open(LISTING).read()
names = sorted(set(re.findall(r"^(_Z\w+):", text, re.M)))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
hits = [(n, d) for n, d in zip(names, dem) if want in d]
if len(hits) != 1:
    print("matches:", [d for _, d in hits][:20])
    sys.exit(1)
name, full = hits[0]
start = text.index("\n" + name + ":")
body = [l.strip() for l in text[start:text.index(".Lfunc_end", start)].splitlines() if l.strip() and not l.strip().startswith(";")]
meta = text[text.index("amdhsa.kernels"):]
blk = meta[max(0, meta.index(".name:           " + name) - 1500):meta.index(".name:           " + name) + 800]
def fig(key):
    m = re.findall(r"\." + key + r":\s+(\d+)", blk)
    return m[-1] if m else "?"


print(full)
print(f"instructions {len(body)}  VGPR {fig('vgpr_count')}  SGPR {fig('sgpr_count')}  scratch {fig('private_segment_fixed_size')} B  "
      f"spilled VGPR {fig('vgpr_spill_count')} / SGPR {fig('sgpr_spill_count')}")
keep = re.compile(r"^(s_load|s_buffer_load|global_load|buffer_load|flat_load|scratch_|ds_write|ds_read|s_waitcnt|s_barrier|s_cbranch|s_branch|\.LBB|v_mfma)")
for i, l in enumerate(body[:first] if "--all" not in sys.argv else body):
    if "--all" in sys.argv or keep.match(l):
        print(f"{i:5d}  {l[:110]}")

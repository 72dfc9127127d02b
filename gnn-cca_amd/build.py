"""Builds libgnncca_mpn.so (HIP kernels + C-ABI) for gfx950 in-tree with hipcc.  No JIT cache: the .so sits next
to the sources so it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgnncca_mpn.so")
SOURCES = ["pack.cpp", "post_host.cpp", "mpn_forward.hip", "graph_build.hip"]
HEADERS = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cuh", ".inc"))] + [os.path.join(ROOT, "include", "gnncca_mpn.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_stamps(verbose=False):
    """Diagnostic twin (tools/stamps.py): same sources with -DGNNCCA_STAMPS -> lib/libgnncca_mpn_stamps.so."""
    return build(force=True, verbose=verbose, out=os.path.join(LIB_DIR, "libgnncca_mpn_stamps.so"), defs=["-DGNNCCA_STAMPS"])


def build_f16_ablations(verbose=False):
    """Diagnostic twin (tools/ab_f16_ablations.sh): the fp16-split encoder GEMM's ablation kernels compiled in -> lib/libgnncca_mpn_f16abl.so
    (select with GNNCCA_DIAG=1 GNNCCA_LIB=... GNNCCA_GEMM_F16_DIAG=n)."""
    return build(force=True, verbose=verbose, out=os.path.join(LIB_DIR, "libgnncca_mpn_f16abl.so"), defs=["-DGNNCCA_F16_ABLATIONS"])


def build(force=False, verbose=False, out=None, defs=()):
    if out is None and not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    out = out or LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function",
           "-fvisibility=hidden", "-DGNNCCA_BUILD",
           # keep MFMA accumulators in VGPRs: the step kernels post-process every accumulator element on the VALU
           # (ReLU + segment sum), and AGPR results would cost one v_accvgpr_read per element
           "-mllvm", "-amdgpu-mfma-vgpr-form"]
    cmd += list(defs)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps(verbose=True))
    elif "--f16-ablations" in sys.argv:
        print(build_f16_ablations(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))

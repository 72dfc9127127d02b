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


def _deps(src_path):
    """The quoted includes of a source, transitively (csrc/ and include/): what its object file has to be newer than."""
    import re
    seen, todo = set(), [src_path]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), flags=re.M):
            for base in (CSRC, os.path.join(ROOT, "include")):
                cand = os.path.join(base, inc)
                if os.path.exists(cand):
                    todo.append(cand)
    return seen


def build(force=False, verbose=False, out=None, defs=()):
    """One object per source (lib/obj/, rebuilt when the source, one of ITS includes or this script is newer), then the link: a change to
    the host-side sources does not recompile the 90-second kernel translation unit."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    out = out or LIB_PATH
    tag = "".join(sorted(defs)).replace("-D", "_") if defs else ""
    obj_dir = os.path.join(LIB_DIR, "obj" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function",
             "-fvisibility=hidden", "-DGNNCCA_BUILD",
             # keep MFMA accumulators in VGPRs: the step kernels post-process every accumulator element on the VALU
             # (ReLU + segment sum), and AGPR results would cost one v_accvgpr_read per element
             "-mllvm", "-amdgpu-mfma-vgpr-form"] + list(defs)
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(d) for d in _deps(sp) | {os.path.abspath(__file__)})
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [_hipcc()] + flags + ["-c", sp, "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), obj))
    failed = False
    for pr, obj in procs:       # the sources compile side by side
        if pr.wait() != 0:
            failed = True
        else:
            os.replace(obj + ".tmp", obj)
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out + ".tmp"]
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

"""Builds reconstructor_amd/librcn.so: hand-written HIP for gfx950 behind the C ABI of include/rcn.h.

In-tree on purpose: the .so travels with the repo snapshot to the GPU box.
build(diag=True) builds tools/librcn_diag.so from the same sources with -DRCN_DIAG: the only
binary in which the ablation / alternative-path environment switches exist (tools/, never the product).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librcn.so")
SO_DIAG = os.path.join(ROOT, "tools", "librcn_diag.so")
SOURCES = ["ctx.hip", "match.hip", "ba.hip", "ba_session.hip", "validity.hip", "fmat.hip", "shard.hip", "store.hip", "desc.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
LIBS = ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-z,defs"]      # -z defs: an unresolved symbol fails the build, not the first dlopen


def _stale(so):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + \
           [os.path.join(ROOT, "include", "rcn.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, diag=False):
    so = SO_DIAG if diag else SO
    if not (force or _stale(so)):
        return so
    objs = []
    procs = []
    extra = ["-DRCN_DIAG"] if diag else []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".diag.o" if diag else ".o"))
        cmd = [HIPCC, *FLAGS, *extra, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs, *LIBS]
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))

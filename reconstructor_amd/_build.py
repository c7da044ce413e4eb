"""Builds reconstructor_amd/librcn.so: hand-written HIP for gfx950 behind the C ABI of include/rcn.h.

In-tree on purpose: the .so travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librcn.so")
SOURCES = ["ctx.hip", "match.hip", "ba.hip", "validity.hip", "fmat.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-munsafe-fp-atomics"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(HERE, "..", "include", "rcn.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not (force or _stale()):
        return SO
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs,
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

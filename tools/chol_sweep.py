"""tools/chol_sweep.py ENV v1 v2 ... [-- n_cams n_points reps] -- A/B of one diagnostic-build switch on the cfg-5 factorisation: one process, one
scene, a fresh ctx per value (rcn_create reads the environment in the diagnostic build), values interleaved over two passes.
Prints the factorisation's time per LM iteration (HIP events inside the solve) per value.
CAUTION (round 4): valid for switches that do not change the number of streams.  The contexts of one process share the runtime's hardware queues;
a switch that adds a stream (a second bulk stream, a stream with another CU mask) measured 13-14 ms per factorisation here and 9.1-10.5 ms in a
process of its own (tools/chol_ab.sh: one process per value) -- use that for such switches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RCN_LIB", os.path.join(ROOT, "tools", "librcn_diag.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from reconstructor_amd import _lib, ba, synth_ba  # noqa: E402

args = sys.argv[1:]
tail = []
if "--" in args:
    k = args.index("--")
    args, tail = args[:k], args[k + 1:]
env, vals = args[0], args[1:]
nc, npts, reps = (int(tail[0]), int(tail[1]), int(tail[2])) if len(tail) >= 3 else (1000, 100000, 3)
sc = synth_ba.make_scene(nc, npts, seed=2024)
res = {v: [] for v in vals}
for rnd in range(2):
    for v in vals:
        os.environ[env] = v
        ctx = _lib.Context(0)
        for i in range(reps):
            P, I, X, s = ba.solve_scene(ctx, sc)
            if i:
                res[v].append((1e3 * s["cholesky_seconds"] / s["iterations"], s["iterations"] / s["solve_seconds"], s["final_rms_px"]))
        del ctx
for v in vals:
    ch = sorted(r[0] for r in res[v])
    print("%s=%-6s chol ms/iter min %.3f median %.3f max %.3f | it/s median %.1f | rms %.9f" %
          (env, v, ch[0], ch[len(ch) // 2], ch[-1], sorted(r[1] for r in res[v])[len(ch) // 2], res[v][0][2]), flush=True)

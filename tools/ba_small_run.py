"""tools/ba_small_run.py [n_cams n_points reps] -- repeated solves of one small scene (the reference's own regime) for rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from reconstructor_amd import _lib, ba, synth_ba
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 25
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = _lib.Context(0)
sc = synth_ba.make_scene(nc, npts, obs_per_point=min(8, nc), seed=5)
other = synth_ba.make_scene(3, 20, obs_per_point=3, seed=6)
ts = []
for i in range(reps):
    ba.solve_scene(ctx, other)           # drops the pair lists: the timed solve builds them, as the reference's loop does
    t0 = time.perf_counter()
    P, I, X, s = ba.solve_scene(ctx, sc)
    ts.append((time.perf_counter() - t0, s["solve_seconds"], s["iterations"]))
ts.sort()
print("%d cams %d pts: %d iterations, solve_seconds median %.3f ms (call %.3f ms), n=%d" % (nc, npts, ts[0][2], 1e3 * ts[len(ts) // 2][1], 1e3 * ts[len(ts) // 2][0], s["reduced_dim"]))

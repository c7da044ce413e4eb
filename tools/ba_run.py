"""tools/ba_run.py -- run one BA config on the GPU (for rocprofv3): python3 tools/ba_run.py 1000 100000"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from reconstructor_amd import _lib, ba, synth_ba
nc, npts = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = _lib.Context(0)
sc = synth_ba.make_scene(nc, npts, seed=2024)
for i in range(reps):
    P, I, X, s = ba.solve_scene(ctx, sc)
    print("run %d: %d iterations %.4f s (%.2f it/s) rms %.6f -> %.6f n=%d" % (i, s["iterations"], s["solve_seconds"], s["iterations"] / s["solve_seconds"], s["initial_rms_px"], s["final_rms_px"], s["reduced_dim"]),
          "| per iteration: schur %.3f chol %.3f tri %.3f ms" % tuple(1e3 * s[k] / max(1, s["iterations"]) for k in ("schur_seconds", "cholesky_seconds", "trisolve_seconds")))

"""tools/fmat_pair_latency.py -- latency of the per-pair plugin path of the epipolar filter (rcn_fmat_filter, host pointers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reconstructor_amd import _lib, fmat, synth_fmat
ctx = _lib.Context(0)
cases = [synth_fmat.two_view(n, 0.3, seed=s) for s, n in enumerate(np.random.default_rng(0).integers(200, 900, 60))]
fmat.estimate_fundamental_inliers(ctx, cases[0][0], cases[0][1])
t0 = time.perf_counter()
for a, b, _ in cases:
    fmat.estimate_fundamental_inliers(ctx, a, b)
print("per-pair epipolar filter: %d calls, %.2f ms per call" % (len(cases), 1e3 * (time.perf_counter() - t0) / len(cases)))

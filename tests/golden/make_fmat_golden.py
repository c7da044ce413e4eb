"""Generates tests/golden/fmat_small.npz: a CSR batch of synthetic two-view match sets and the
expected inlier masks / counts / iteration counts of the epipolar filter, from the oracle.  Pairs with
7, 14 or >= 15 points are cross-checked against the independent numpy transcription (tests/indep.py)
before writing; for 8..13 points OpenCV's LMedS picks among exact fits by rounding noise, so only
count and iterations are compared there.  Run from the repo root."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import indep  # noqa: E402
from oracle import orc_fmat  # noqa: E402
from reconstructor_amd import synth_fmat  # noqa: E402

sizes = [0, 5, 7, 8, 10, 13, 14, 15, 16, 21, 33, 64, 100, 180, 257, 400, 640] + list(np.random.default_rng(3).integers(15, 300, 15))
off, a, b = synth_fmat.grid(sizes, 0.35, seed=21)
mask, counts, iters = orc_fmat.filter_grid(off, a, b, threads=4)
for p, n in enumerate(sizes):
    m, c, it = indep.fmat_python(a[off[p]:off[p + 1]], b[off[p]:off[p + 1]])
    assert c == counts[p] and it == iters[p], (p, n, c, counts[p], it, iters[p])
    if not 8 <= n <= 13:
        assert (m == mask[off[p]:off[p + 1]]).all(), (p, n)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "fmat_small.npz"), pair_off=off, xy1=a, xy2=b,
                    mask=mask, counts=counts, iterations=iters)
print("pairs %d, points %d, inliers %d" % (len(sizes), off[-1], mask.sum()))

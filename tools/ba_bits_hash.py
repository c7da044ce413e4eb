"""tools/ba_bits_hash.py -- one SHA-256 over the results of five solves (cfg 5, cfg 4, 25 / 6 / 3 cameras): a change that must not move a bit prints the same line
(round 5, before the change to k_ba_eval's staging: 8feb94621a42923c1bb83bda78219f55980d898d52a1b1a5eb975c2572cc8c3a)"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import numpy as np
from reconstructor_amd import _lib, ba, synth_ba
ctx = _lib.Context(0)
h = hashlib.sha256()
for nc, npts, opp in ((1000, 100000, 10), (200, 20000, 10), (25, 1500, 8), (6, 300, 6), (3, 40, 3)):
    sc = synth_ba.make_scene(nc, npts, obs_per_point=min(opp, nc), seed=77)
    P, I, X, s = ba.solve_scene(ctx, sc)
    for a in (P, I, X):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(repr((s["iterations"], s["final_cost"])).encode())
print(h.hexdigest())

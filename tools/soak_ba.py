#!/usr/bin/env python3
"""tools/soak_ba.py [seconds] [seed] -- randomised differential run of the GPU bundle adjustment against the CPU oracle:
random camera / landmark / track counts on both sides of the 10-camera switch (constant vs bounded intrinsics), easy and
far-off starts (the latter run the projected line search and rejected steps).  Reports how many solves agree on
(iterations, termination, line-search backtracks) and the largest RMS difference among those; a solve whose iteration
counts differ is listed (long, badly conditioned runs may legitimately part ways: 1e-16 differences grow along 50 steps).
Not part of the test-suite; run on the MI355X box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from oracle import orc_ba
    from reconstructor_amd import _lib, ba, synth_ba
    ctx = _lib.Context(0)
    rng = np.random.default_rng(seed)
    t0, parted, bt = time.time(), [], 0
    stat = {False: {"n": 0, "agree": 0, "close": 0, "worst_close": 0.0}, True: {"n": 0, "agree": 0, "close": 0, "worst_close": 0.0}}
    while time.time() - t0 < budget:
        nc = int(rng.integers(3, 15))
        npts = int(rng.integers(20, 400))
        k = int(rng.integers(2, min(nc, 8) + 1))
        hard = rng.random() < 0.4
        pert = (0.4, 2.0, 1.5) if hard else (0.01, 0.05, 0.05)
        sc = synth_ba.make_scene(nc, npts, obs_per_point=k, seed=int(rng.integers(1 << 30)), perturb=pert,
                                 focal_factor=float(rng.choice([1.2, 1.2, 1.9, 2.2])))
        P0, I0, X0, s0 = orc_ba.solve(sc, threads=8)
        P1, I1, X1, s1 = ba.solve_scene(ctx, sc)
        bt += s0["line_search_backtracks"]
        key = lambda s: (s["iterations"], s["termination"], s["line_search_backtracks"], s["successful_steps"])
        st = stat[hard]
        st["n"] += 1
        d = abs(s0["final_rms_px"] - s1["final_rms_px"])
        if key(s0) == key(s1):
            st["agree"] += 1
            if d <= 1e-5:
                st["close"] += 1
                st["worst_close"] = max(st["worst_close"], d)
        if key(s0) != key(s1) or d > 1e-5:
            parted.append((nc, npts, k, hard, key(s0), key(s1), s0["final_rms_px"], s1["final_rms_px"]))
    for hard in (False, True):
        st = stat[hard]
        print("%s starts: %d solves; %d agree on (iterations, termination, backtracks, accepted steps), %d of them with |RMS difference| <= 1e-5 px "
              "(largest %.2g)" % ("far-off" if hard else "near", st["n"], st["agree"], st["close"], st["worst_close"]))
    print("%d oracle line-search backtracks in total; %.0f s, seed %d" % (bt, time.time() - t0, seed))
    for p in parted[:12]:
        print("  parted:", p)


if __name__ == "__main__":
    main()

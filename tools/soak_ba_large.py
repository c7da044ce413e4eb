#!/usr/bin/env python3
"""tools/soak_ba_large.py [seconds] [seed] [min_cams max_cams] -- bundle adjustment at sizes where the factorisation runs its
three-stream schedule (default 40 .. 300 cameras: reduced systems of 3 .. 24 blocks of 128; from 27 blocks, ~340 cameras, on the
two-panel bulk updates with their tile-level hand-off take part): every random problem is solved twice on the GPU
and the two results must be equal bit for bit (a race in the stream hand-offs would show as a difference or a failed
step); every fifth is also solved by the CPU oracle and must agree on the iteration count and on the final RMS to 1e-5 px.
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
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (40, 300)
    from oracle import orc_ba
    from reconstructor_amd import _lib, ba, synth_ba
    ctx = _lib.Context(0)
    rng = np.random.default_rng(seed)
    t0, n, checked, blocks = time.time(), 0, 0, set()
    t_said = t0
    while time.time() - t0 < budget:
        nc = int(rng.integers(lo, hi))
        npts = int(rng.integers(8, 30)) * nc
        k = int(rng.integers(3, 9))
        sc = synth_ba.make_scene(nc, npts, obs_per_point=k, seed=int(rng.integers(1 << 30)))
        P1, I1, X1, s1 = ba.solve_scene(ctx, sc)
        P2, I2, X2, s2 = ba.solve_scene(ctx, sc)
        same = (P1.tobytes() == P2.tobytes() and I1.tobytes() == I2.tobytes() and X1.tobytes() == X2.tobytes()
                and s1["iterations"] == s2["iterations"] and s1["final_cost"] == s2["final_cost"])
        if not same or s1["invalid_steps"] or s1["final_rms_px"] > 1.0:
            print("FAILED", nc, npts, k, s1["iterations"], s2["iterations"], s1["final_rms_px"], s2["final_rms_px"], s1["invalid_steps"], flush=True)
            sys.exit(1)
        blocks.add((s1["reduced_dim"] + 1 + 127) // 128)
        if n % 5 == 0:
            P0, I0, X0, s0 = orc_ba.solve(sc, threads=16)
            if s0["iterations"] != s1["iterations"] or abs(s0["final_rms_px"] - s1["final_rms_px"]) > 1e-5:
                print("ORACLE DIFFERS", nc, npts, k, s0["iterations"], s1["iterations"], s0["final_rms_px"], s1["final_rms_px"], flush=True)
                sys.exit(1)
            checked += 1
        n += 1
        if time.time() - t_said > 60.0:      # (a run that says nothing for minutes is taken for hung)
            t_said = time.time()
            print("  ... %d problems so far, %.0f s" % (n, time.time() - t0), flush=True)
    print("soak ok: %d problems solved twice, bit-equal; %d of them equal to the oracle; factorisations of %d .. %d blocks; %.0f s, seed %d"
          % (n, checked, min(blocks), max(blocks), time.time() - t0, seed))


if __name__ == "__main__":
    main()

"""Writes tests/golden/ba_cfg5.npz: the CPU oracle's solve of BASELINE.json configs[4] (1000 cameras, 100 000
points, 1 000 000 observations; synth_ba.make_scene(1000, 100000, obs_per_point=10, seed=2024) -- the scene
bench.py times) and, with --cfg4, of configs[3] into ba_cfg4.npz.

  python tests/golden/make_ba_cfg5_golden.py [--cfg4] [threads]     (cfg 5: tens of minutes on 8 cores)

Stored: iteration count, termination, step counts, cost trace, initial / final cost and RMS, and a sample of the
adjusted parameters (every 10th camera, every 500th point) for a direct comparison with the GPU solve.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc_ba                    # noqa: E402
from reconstructor_amd import synth_ba      # noqa: E402


def main():
    cfg4 = "--cfg4" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    threads = int(args[0]) if args else 8
    nc, npts = (200, 20000) if cfg4 else (1000, 100000)
    sc = synth_ba.make_scene(nc, npts, obs_per_point=10, seed=2024)
    t0 = time.time()
    P, I, X, s = orc_ba.solve(sc, threads=threads)
    dt = time.time() - t0
    path = os.path.join(ROOT, "tests", "golden", "ba_cfg4.npz" if cfg4 else "ba_cfg5.npz")
    np.savez_compressed(path, args=np.array([nc, npts, 10, 2024]), iterations=s["iterations"], termination=s["termination"],
                        successful_steps=s["successful_steps"], unsuccessful_steps=s["unsuccessful_steps"],
                        cost_trace=s["cost_trace"], initial_cost=s["initial_cost"], final_cost=s["final_cost"],
                        initial_rms_px=s["initial_rms_px"], final_rms_px=s["final_rms_px"], reduced_dim=s["reduced_dim"],
                        poses_sample=P[::10], intrinsics_sample=I[::10], points_sample=X[::500],
                        oracle_seconds=dt, oracle_threads=threads)
    print(path, "%d iterations, termination %d, rms %.9f -> %.9f px, %.1f s on %d threads"
          % (s["iterations"], s["termination"], s["initial_rms_px"], s["final_rms_px"], dt, threads))


if __name__ == "__main__":
    main()

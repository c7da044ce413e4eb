"""Generates tests/golden/ba_*.npz: residual/Jacobian spot values and small-scene solve traces.

The reference holds no BA fixtures.  Expected values come from the C
oracle (oracle/ba_oracle.c) and are cross-checked here before being written:
  * residuals against an independent numpy projection (reconstructor_amd/synth_ba.project),
  * Jacobians against central differences,
  * the converged cost of each scene against scipy.optimize.least_squares on the same free
    parameters (same gauge: camera 0 fixed, camera 1 translation fixed, intrinsics per mode).
Run from the repo root:  python tests/golden/make_ba_golden.py
"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc_ba  # noqa: E402
from reconstructor_amd import synth_ba  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def free_mask(nc, mode):
    m = np.zeros((nc, 12), bool)
    m[1:, :3] = True
    m[2:, 3:6] = True
    if mode == 1:
        m[:, [6, 7, 10, 11]] = True
    return m


def scipy_min_cost(sc, mode):
    nc, npts = sc["poses"].shape[0], sc["points"].shape[0]
    cam0 = np.concatenate([sc["poses"], sc["intrinsics"]], 1)
    mask = free_mask(nc, mode)

    def unpack(x):
        cam = cam0.copy()
        cam[mask] = x[:mask.sum()]
        return cam, x[mask.sum():].reshape(npts, 3)

    def fun(x):
        cam, pts = unpack(x)
        uv, _ = synth_ba.project(cam[sc["obs_cam"], :6], cam[sc["obs_cam"], 6:], pts[sc["obs_pt"]])
        return (uv - sc["obs_uv"]).ravel()

    x0 = np.concatenate([cam0[mask], sc["points"].ravel()])
    res = least_squares(fun, x0, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=200)
    return res.cost


def main():
    rng = np.random.default_rng(17)
    spots = {"pose": [], "intr": [], "X": [], "uv": [], "res": [], "J": []}
    for t in range(10):
        scale = [1e-9, 1e-4, 5e-3, 0.3, 1.0, 2.0, 3.0, 0.02, 0.7, 1.5][t]   # includes theta ~ 0
        pose = np.concatenate([rng.standard_normal(3) * scale, rng.standard_normal(3) + [0, 0, 8]])
        intr = np.array([600 + 20 * rng.standard_normal(), 600 + 20 * rng.standard_normal(), 256, 168,
                         0.02 * rng.standard_normal(), 0.005 * rng.standard_normal()])
        X = rng.standard_normal(3)
        uv = np.trunc(rng.random(2) * 400)
        res, J = orc_ba.residual_jacobian(pose, intr, X, uv)
        pu, _ = synth_ba.project(pose, intr, X)
        assert np.allclose(res, pu - uv, rtol=0, atol=1e-9)
        x = np.concatenate([pose, intr, X])
        for k in range(15):
            h = 1e-6 * max(1.0, abs(x[k]))
            xp, xm = x.copy(), x.copy()
            xp[k] += h; xm[k] -= h
            rp, _ = orc_ba.residual_jacobian(xp[:6], xp[6:12], xp[12:], uv, False)
            rm, _ = orc_ba.residual_jacobian(xm[:6], xm[6:12], xm[12:], uv, False)
            assert np.allclose(J[:, k], (rp - rm) / (2 * h), rtol=2e-6, atol=2e-6 * np.abs(J).max()), (t, k)
        for k, v in zip(spots, (pose, intr, X, uv, res, J)):
            spots[k].append(v)
    np.savez_compressed(os.path.join(OUT, "ba_spots.npz"), **{k: np.array(v) for k, v in spots.items()})
    print("spots: 10 residual/Jacobian triples (theta from 1e-9 to 3)")

    scenes = {}
    for name, nc, npts, k, seed in (("cams3", 3, 20, 3, 5), ("cams12", 12, 300, 10, 5)):
        sc = synth_ba.make_scene(nc, npts, obs_per_point=k, seed=seed)
        P, I, X, s = orc_ba.solve(sc, threads=2)
        mode = 0 if nc < 10 else 1
        cmin = scipy_min_cost(sc, mode)
        # LM stops on the 1e-6 function tolerance: its cost is within ~1e-6 (relative) of the minimum
        assert cmin <= s["final_cost"] * (1 + 1e-9) and (s["final_cost"] - cmin) <= 5e-6 * cmin, (s["final_cost"], cmin)
        scenes.update({name + "/args": np.array([nc, npts, k, seed]), name + "/cost_trace": s["cost_trace"],
                       name + "/final_rms_px": s["final_rms_px"], name + "/iterations": s["iterations"],
                       name + "/termination": s["termination"], name + "/scipy_min_cost": cmin,
                       name + "/poses": P, name + "/intrinsics": I, name + "/points": X})
        print("%s: %d iterations, rms %.6f px, cost %.9g (scipy minimum %.9g)" % (name, s["iterations"], s["final_rms_px"], s["final_cost"], cmin))
    np.savez_compressed(os.path.join(OUT, "ba_scenes.npz"), **scenes)


if __name__ == "__main__":
    main()

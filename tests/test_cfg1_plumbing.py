"""BASELINE configs[0] ("Fountain 25-img set ... all-pairs match + incremental BA", plumbing):
the Fountain images cannot be processed here (no OpenCV), so the same plumbing runs on a
synthetic 25-image set: SIFT-like 128-d descriptors through the whole pair grid, and the
reference's incremental loop -- a fresh GLOBAL bundle adjustment after every registered view
(SequentialReconstructor.cpp:1040-1094), 3 -> 25 cameras, state carried from solve to solve."""
import numpy as np
import pytest

from oracle import orc, orc_ba
from reconstructor_amd import synth, synth_ba


def _sub_problem(sc, poses, intr, pts, n):
    """Cameras 0..n-1 (imgIdxOrder grows by one per view, :810) and the landmarks seen by >= 2 of them."""
    keep_obs = sc["obs_cam"] < n
    cnt = np.bincount(sc["obs_pt"][keep_obs], minlength=pts.shape[0])
    live = cnt >= 2
    keep_obs &= live[sc["obs_pt"]]
    remap = np.cumsum(live) - 1
    return {"poses": poses[:n].copy(), "intrinsics": intr[:n].copy(), "points": pts[live].copy(),
            "obs_uv": np.ascontiguousarray(sc["obs_uv"][keep_obs]),
            "obs_cam": np.ascontiguousarray(sc["obs_cam"][keep_obs]),
            "obs_pt": np.ascontiguousarray(remap[sc["obs_pt"][keep_obs]].astype(np.int32))}, live


def _incremental(solve, sc, n_cams):
    poses, intr, pts = sc["poses"].copy(), sc["intrinsics"].copy(), sc["points"].copy()
    trace = []
    for n in range(3, n_cams + 1):
        sub, live = _sub_problem(sc, poses, intr, pts, n)
        P, I, X, s = solve(sub)
        poses[:n], intr[:n], pts[live] = P, I, X          # adjust() updates the maps in place
        trace.append((n, s["iterations"], s["termination"], s["final_rms_px"], s["reduced_dim"]))
    return poses, intr, pts, trace


def test_incremental_ba_oracle_runs_and_switches_branch():
    """CPU tier: the oracle walks the whole loop; the intrinsics branch flips at 10 cameras."""
    sc = synth_ba.make_scene(25, 1500, obs_per_point=6, seed=31)
    _, _, _, trace = _incremental(lambda sub: orc_ba.solve(sub, threads=4), sc, 25)
    assert [t[0] for t in trace] == list(range(3, 26))
    for n, it, term, rms, dim in trace:
        assert dim == (6 * (n - 1) - 3 if n < 10 else 6 * (n - 1) - 3 + 4 * n)   # BundleAdjuster.cpp:112-121
        assert rms < 1.0 and term in (1, 2, 3)


@pytest.mark.gpu
def test_incremental_ba_gpu_matches_oracle(gpu_ctx):
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(25, 1500, obs_per_point=6, seed=31)
    P0, I0, X0, t0 = _incremental(lambda sub: orc_ba.solve(sub, threads=4), sc, 25)
    P1, I1, X1, t1 = _incremental(lambda sub: ba.solve_scene(gpu_ctx, sub), sc, 25)
    for a, b in zip(t0, t1):
        assert a[:3] == b[:3] and a[4] == b[4], (a, b)
        assert abs(a[3] - b[3]) <= 1e-5
    assert np.allclose(P1, P0, atol=1e-6) and np.allclose(X1, X0, atol=1e-5) and np.allclose(I1, I0, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_grid_25_images_sift_like(gpu_ctx):
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs, match_features_grid
    ks = [1400 + 8 * ((7 * i) % 25) for i in range(25)]
    ims = synth.descriptor_set("sift", 25, ks, n_world=5000, seed=19)
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    fm = match_features_grid(m, ims)                         # SequentialReconstructor::matchFeatures, no filter
    pairs = all_pairs(25)
    exp, counts = orc.match_grid(ims, pairs, threads=0)
    assert len(pairs) == 300
    for p, (a, b) in enumerate(pairs):
        fwd = {int(q): int(exp[p, q]) for q in np.nonzero(exp[p, :ks[a]] >= 0)[0]}
        assert fm[(a, b)] == fwd
        assert fm[(b, a)] == {t: q for q, t in fwd.items()}   # inverse pair = inverted map (:219-227)
    m.clear()

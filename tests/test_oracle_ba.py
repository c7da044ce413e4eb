"""CPU suite: the BA oracle against the committed golden vectors (no GPU)."""
import os

import numpy as np

from oracle import orc_ba
from reconstructor_amd import synth_ba

G = os.path.join(os.path.dirname(__file__), "golden")


def test_residual_jacobian_spots():
    z = np.load(os.path.join(G, "ba_spots.npz"))
    for i in range(len(z["pose"])):
        res, J = orc_ba.residual_jacobian(z["pose"][i], z["intr"][i], z["X"][i], z["uv"][i])
        assert np.allclose(res, z["res"][i], rtol=1e-13, atol=1e-12)
        assert np.allclose(J, z["J"][i], rtol=1e-12, atol=1e-12)
        # additive distortion (BundleAdjuster.h:43-47): the SAME scalar is added to x and y
        pu, _ = synth_ba.project(z["pose"][i], z["intr"][i], z["X"][i])
        assert np.allclose(res, pu - z["uv"][i], atol=1e-9)


def test_scene_traces_match_golden_and_scipy_minimum():
    z = np.load(os.path.join(G, "ba_scenes.npz"))
    for name in ("cams3", "cams12"):
        nc, npts, k, seed = z[name + "/args"]
        sc = synth_ba.make_scene(int(nc), int(npts), obs_per_point=int(k), seed=int(seed))
        P, I, X, s = orc_ba.solve(sc, threads=2)
        assert s["iterations"] == int(z[name + "/iterations"]) and s["termination"] == int(z[name + "/termination"])
        assert np.allclose(s["cost_trace"], z[name + "/cost_trace"], rtol=1e-9)
        assert abs(s["final_rms_px"] - float(z[name + "/final_rms_px"])) < 1e-9
        assert np.allclose(P, z[name + "/poses"], atol=1e-9) and np.allclose(X, z[name + "/points"], atol=1e-8)
        cmin = float(z[name + "/scipy_min_cost"])
        assert cmin <= s["final_cost"] * (1 + 1e-9) and s["final_cost"] - cmin <= 5e-6 * cmin


def test_gauge_and_intrinsics_branches():
    # < 10 cameras: intrinsics constant (BundleAdjuster.cpp:112-115); camera 0 pose and camera 1
    # translation never move (:100-105)
    sc = synth_ba.make_scene(6, 80, obs_per_point=5, seed=1)
    P, I, X, s = orc_ba.solve(sc)
    assert np.array_equal(I, sc["intrinsics"])
    assert np.array_equal(P[0], sc["poses"][0]) and np.array_equal(P[1, 3:], sc["poses"][1, 3:])
    assert not np.array_equal(P[1, :3], sc["poses"][1, :3])
    assert s["reduced_dim"] == 3 + 6 * 4
    # >= 10 cameras: fx, fy, k1, k2 free; cx, cy constant (:117-121)
    sc = synth_ba.make_scene(10, 200, obs_per_point=6, seed=2)
    P, I, X, s = orc_ba.solve(sc)
    assert np.array_equal(I[:, 2:4], sc["intrinsics"][:, 2:4]) and not np.array_equal(I[:, 0], sc["intrinsics"][:, 0])
    assert s["reduced_dim"] == 3 + 6 * 8 + 4 * 10
    assert (I[:, :2] <= 1000.0).all()


def test_upper_bound_is_a_projection():
    # focal lengths that start above the 1000 bound are clamped by the first Plus (:120-121)
    sc = synth_ba.make_scene(10, 150, obs_per_point=6, seed=4, focal_factor=2.2)   # f = 1126 > 1000
    P, I, X, s = orc_ba.solve(sc)
    assert (I[:, :2] <= 1000.0 + 1e-12).all() and s["bound_projections"] > 0


def test_noise_free_converges_to_zero():
    sc = synth_ba.make_scene(12, 200, seed=8, noise_px=0.0, integer_obs=False)
    P, I, X, s = orc_ba.solve(sc)
    assert s["final_rms_px"] < 1e-6
    assert np.allclose(X, sc["points_gt"], atol=1e-5)


def shifted_principal_point(sc, shift):
    """The same optimisation problem with cx, cy and every observation moved by `shift`: residuals and Jacobians
    are unchanged, only the value of a parameter block that is CONSTANT for < 10 cameras differs."""
    sc2 = {k: np.array(v, copy=True) for k, v in sc.items()}
    sc2["intrinsics"][:, 2:4] += shift
    sc2["obs_uv"] = sc2["obs_uv"] + shift
    return sc2


def test_parameter_tolerance_ignores_blocks_outside_the_reduced_program():
    """Ceres takes |x| of the parameter-tolerance test over the reduced program: constant blocks (all intrinsics
    when there are < 10 cameras, camera 0's pose) have been removed from it.  Moving the principal point by 1e6 px
    together with the observations leaves the problem unchanged -- so it must leave the termination unchanged; with
    |x| taken over everything the shifted problem stops after its first iteration."""
    sc = synth_ba.make_scene(6, 120, obs_per_point=5, seed=11)
    o = orc_ba.default_options(6)
    o.parameter_tolerance = 1e-7
    o.function_tolerance = 0.0            # let the parameter test decide
    P0, I0, X0, s0 = orc_ba.solve(sc, o)
    P1, I1, X1, s1 = orc_ba.solve(shifted_principal_point(sc, 1.0e6), o)
    assert s0["termination"] == 3 == s1["termination"]                       # RCN_BA_CONVERGENCE_PARAMETER
    assert s0["iterations"] == s1["iterations"] and s0["iterations"] >= 3
    assert np.allclose(s0["cost_trace"], s1["cost_trace"], rtol=1e-6)


def _next_step_numpy(start, prev, cur, lo, hi):
    """InterpolatingPolynomialMinimizingStepSize written with numpy's own solver and root finder."""
    if not cur[3]:
        return min(max(cur[0] * 0.5, lo), hi)
    S = [start, cur] + ([prev] if prev[3] else [])
    deg = sum(int(s[3]) + int(s[4]) for s in S) - 1
    A, b = [], []
    for s in S:
        if s[3]:
            A.append([s[0] ** (deg - j) for j in range(deg + 1)]); b.append(s[1])
        if s[4]:
            A.append([(deg - j) * s[0] ** (deg - j - 1) if j < deg else 0.0 for j in range(deg + 1)]); b.append(s[2])
    c = np.linalg.solve(np.array(A), np.array(b))
    best_x = (lo + hi) / 2
    best = np.polyval(c, best_x)
    for x in (lo, hi):
        if np.polyval(c, x) < best:
            best, best_x = np.polyval(c, x), x
    for r in np.roots(np.polyder(c)):
        if lo <= r.real <= hi and np.polyval(c, r.real) < best:
            best, best_x = np.polyval(c, r.real), r.real
    return best_x


def test_line_search_polynomials_against_numpy():
    """Cubic (two samples) and quintic (three samples) interpolants, samples without a valid gradient, and the
    no-valid-value bisection: the oracle's elimination / root iteration against numpy's LAPACK paths."""
    rng = np.random.default_rng(0)
    for _ in range(1500):
        f0, g0 = rng.uniform(1, 100), -rng.uniform(0.1, 50)
        x1 = rng.uniform(0.05, 1.0)
        cur = [x1, f0 + rng.uniform(-0.5, 5) * abs(g0) * x1, rng.normal() * 30, 1, rng.random() > 0.1]
        prev = [0, 0, 0, 0, 0]
        if rng.random() < 0.5:
            x2 = x1 / rng.uniform(0.2, 0.9)
            prev = [x2, f0 + rng.uniform(0, 9) * abs(g0) * x2, rng.normal() * 30, 1, rng.random() > 0.1]
        lo, hi = 1e-3 * x1, 0.6 * x1
        got = orc_ba.ls_next_step([0, f0, g0, 1, 1], prev, cur, lo, hi)
        exp = _next_step_numpy([0, f0, g0, 1, 1], prev, cur, lo, hi)
        assert lo <= got <= hi and abs(got - exp) <= 1e-9 * exp
    assert orc_ba.ls_next_step([0, 5, -1, 1, 1], [0, 0, 0, 0, 0], [0.8, np.inf, 0, 0, 0], 8e-4, 0.48) == 0.4


def test_bounds_line_search_backtracks_and_still_descends():
    """Far-off starts with bounded focal lengths: the Armijo search runs (8 / 11 / 10 backtracks), every accepted
    step lowers the cost and the solves end at the noise floor."""
    for seed, backtracks in ((1, 8), (7, 11), (10, 10)):
        sc = synth_ba.make_scene(12, 200, obs_per_point=6, seed=seed, perturb=(0.4, 2.0, 1.5))
        P, I, X, s = orc_ba.solve(sc, threads=4)
        assert s["line_search_backtracks"] == backtracks and s["termination"] == 1
        tr = np.array(s["cost_trace"][: s["iterations"] + 1])
        assert (np.diff(tr) <= 0).all() and s["final_rms_px"] < 0.7
        assert (I[:, :2] <= 1000.0).all()


def test_consecutive_invalid_steps_end_the_solve_at_the_limit_not_one_later():
    """Ceres' HandleInvalidStep: `++num_consecutive_invalid_steps_ >= max_num_consecutive_invalid_steps` -> FAILURE (restated from
    memory; DESIGN.md section 1 records the choice).  On a scene whose every step is invalid (one landmark block exactly zero, no LM
    floor) the solve must take exactly `max` steps, all invalid, and leave the parameters alone."""
    sc = synth_ba.make_always_invalid_scene()
    for limit in (1, 5, 6):
        o = orc_ba.default_options(6)
        o.min_lm_diagonal = 0.0
        o.max_consecutive_invalid_steps = limit
        P, I, X, s = orc_ba.solve(sc, options=o, threads=2)
        assert s["termination"] == 6 and s["invalid_steps"] == limit and s["iterations"] == limit, s
        assert s["successful_steps"] == 0 and s["final_cost"] == s["initial_cost"]
        assert np.array_equal(P, sc["poses"]) and np.array_equal(X, sc["points"])
    # with the floor of the LM diagonal in place (Ceres' default 1e-6) the same scene is solvable: the zero block is damped
    o = orc_ba.default_options(6)
    P, I, X, s = orc_ba.solve(sc, options=o, threads=2)
    assert s["termination"] != 6 and s["invalid_steps"] == 0 and s["final_cost"] < s["initial_cost"]

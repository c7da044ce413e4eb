"""GPU parity suite for bundle adjustment: HIP path (through the C ABI) vs the CPU oracle.
Tolerance from BASELINE.json north_star: final reprojection RMS within 1e-5 px of the CPU path."""
import numpy as np
import pytest

from oracle import orc_ba
from reconstructor_amd import synth_ba

pytestmark = pytest.mark.gpu
RMS_TOL_PX = 1e-5


# (3, 6000, 3): pair segments of 6000 entries (bitonic sort in global memory); (6, 3000, 5) and (25, 2500, 8):
# segments of hundreds (LDS bitonic sort, workgroup-per-block Schur kernel) -- the reference's own regime
@pytest.mark.parametrize("nc,npts,k", [(3, 20, 3), (9, 150, 6), (12, 300, 10), (40, 3000, 10),
                                       (3, 6000, 3), (6, 3000, 5), (25, 2500, 8)])
def test_small_scenes_match_oracle(gpu_ctx, nc, npts, k):
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(nc, npts, obs_per_point=k, seed=5)
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=4)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    print(nc, npts, "oracle", s0["iterations"], s0["final_rms_px"], "gpu", s1["iterations"], s1["final_rms_px"])
    assert abs(s1["initial_cost"] - s0["initial_cost"]) <= 1e-9 * s0["initial_cost"]
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= RMS_TOL_PX
    assert s1["iterations"] == s0["iterations"] and s1["termination"] == s0["termination"]
    n = min(len(s0["cost_trace"]), len(s1["cost_trace"]))
    assert np.allclose(s1["cost_trace"][:n], s0["cost_trace"][:n], rtol=1e-7)
    assert np.allclose(P1, P0, atol=1e-6) and np.allclose(X1, X0, atol=1e-5) and np.allclose(I1, I0, rtol=1e-6, atol=1e-6)
    assert s1["line_search_backtracks"] == 0 == s0["line_search_backtracks"]


def test_cfg4_parity(gpu_ctx):
    """BASELINE cfg 4: 200 cams / 20k points / 200k observations."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(200, 20000, obs_per_point=10, seed=2024)
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=0)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    print("cfg4 oracle", s0["iterations"], s0["final_rms_px"], "%.2fs" % s0["solve_seconds"],
          "gpu", s1["iterations"], s1["final_rms_px"], "%.3fs" % s1["solve_seconds"], "n", s1["reduced_dim"])
    assert s1["reduced_dim"] == 6 * 199 - 3 + 4 * 200
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= RMS_TOL_PX
    assert s1["final_rms_px"] < 0.8 and s1["iterations"] == s0["iterations"]


def test_noise_free_scene_converges_to_zero(gpu_ctx):
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(15, 400, seed=9, noise_px=0.0, integer_obs=False)
    P, I, X, s = ba.solve_scene(gpu_ctx, sc)
    assert s["final_rms_px"] < 1e-6, s


def test_adjust_signature(gpu_ctx):
    """BundleAdjuster::adjust call shape: maps keyed by image index, landmarks with
    triangulatedFeatures, in-place update, returns global->local camera map."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(5, 60, obs_per_point=4, seed=3)
    order = [7, 3, 11, 5, 2]                       # imgIdxOrder: local index = position
    feats = {g: [] for g in order}
    landmarks = []
    for j in range(60):
        tf = []
        for o in np.nonzero(sc["obs_pt"] == j)[0]:
            g = order[sc["obs_cam"][o]]
            feats[g].append((int(sc["obs_uv"][o, 0]), int(sc["obs_uv"][o, 1])))
            tf.append((g, len(feats[g]) - 1))
        landmarks.append({"x": sc["points"][j, 0], "y": sc["points"][j, 1], "z": sc["points"][j, 2],
                          "triangulatedFeatures": tf})
    poses = {}
    for l, g in enumerate(order):
        T = np.eye(4); T[:3, :3] = synth_ba.rodrigues(sc["poses"][l, :3]); T[:3, 3] = sc["poses"][l, 3:]
        poses[g] = T
    intr = {g: sc["intrinsics"][l].copy() for l, g in enumerate(order)}
    adj = ba.BundleAdjuster(ctx=gpu_ctx)
    g2l = adj.adjust(feats, landmarks, poses, intr, order)
    assert g2l == {g: l for l, g in enumerate(order)}
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=2)
    assert abs(adj.last_summary["final_rms_px"] - s0["final_rms_px"]) <= RMS_TOL_PX
    assert np.allclose([[lm["x"], lm["y"], lm["z"]] for lm in landmarks], X0, atol=1e-5)
    assert np.allclose(poses[order[0]], np.vstack([np.c_[synth_ba.rodrigues(sc["poses"][0, :3]), sc["poses"][0, 3:]], [0, 0, 0, 1]]), atol=1e-5)


def test_bounds_active_matches_oracle(gpu_ctx):
    """fx, fy start above the 1000 bound (BundleAdjuster.cpp:120-121): initial projection and
    clamped Plus must follow the oracle step for step."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(10, 150, obs_per_point=6, seed=4, focal_factor=2.2)
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=2)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert (I1[:, :2] <= 1000.0 + 1e-12).all()
    assert s1["iterations"] == s0["iterations"] and s1["termination"] == s0["termination"]
    assert s1["line_search_backtracks"] == s0["line_search_backtracks"]
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= 1e-5


@pytest.mark.parametrize("seed,backtracks", [(1, 8), (7, 11), (10, 10)])
def test_bounds_line_search_follows_the_oracle(gpu_ctx, seed, backtracks):
    """Far-off starts with >= 10 cameras (focal lengths bounded, BundleAdjuster.cpp:117-121): the full LM step
    fails the sufficient-decrease test several times and the projected Armijo search backtracks through the
    cubic / quintic interpolants -- same trial steps, same accepted points as the CPU oracle."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(12, 200, obs_per_point=6, seed=seed, perturb=(0.4, 2.0, 1.5))
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=4)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    print(seed, "oracle", s0["iterations"], s0["line_search_backtracks"], s0["final_rms_px"],
          "gpu", s1["iterations"], s1["line_search_backtracks"], s1["final_rms_px"])
    assert s0["line_search_backtracks"] == backtracks          # the case does what it is here for
    assert s1["line_search_backtracks"] == s0["line_search_backtracks"]
    assert s1["iterations"] == s0["iterations"] and s1["termination"] == s0["termination"]
    n = min(len(s0["cost_trace"]), len(s1["cost_trace"]))
    assert np.allclose(s1["cost_trace"][:n], s0["cost_trace"][:n], rtol=1e-6)
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= RMS_TOL_PX
    assert np.allclose(P1, P0, atol=1e-5) and np.allclose(X1, X0, atol=1e-4)


def test_cfg5_properties(gpu_ctx):
    """BASELINE cfg 5 (1000 cams / 100k points / 1M observations): too big for the CPU oracle
    inside the suite, so size-independent properties: monotone cost trace, gauge untouched,
    reduced dimension 9991, converged RMS at the noise level, bit-reproducible first step."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(1000, 100000, obs_per_point=10, seed=2024)
    P, I, X, s = ba.solve_scene(gpu_ctx, sc)
    print("cfg5", s["iterations"], s["initial_rms_px"], s["final_rms_px"], "%.3fs" % s["solve_seconds"], s["termination"])
    assert s["reduced_dim"] == 9991
    tr = s["cost_trace"]
    assert (np.diff(tr) <= 1e-9 * tr[:-1]).all()
    assert s["final_rms_px"] < 0.8 < s["initial_rms_px"]
    assert np.array_equal(P[0], sc["poses"][0]) and np.array_equal(P[1, 3:], sc["poses"][1, 3:])
    assert np.array_equal(I[:, 2:4], sc["intrinsics"][:, 2:4])
    assert np.abs(X - sc["points_gt"]).mean() < 0.05
    # ---- and against the CPU oracle's own solve of this scene (tests/golden/ba_cfg5.npz, written once in the
    #      build container by tests/golden/make_ba_cfg5_golden.py: the oracle needs ~3e11 flop per iteration here)
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_cfg5.npz"))
    assert list(g["args"]) == [1000, 100000, 10, 2024]
    assert s["iterations"] == int(g["iterations"]) and s["termination"] == int(g["termination"])
    assert s["successful_steps"] == int(g["successful_steps"]) and s["unsuccessful_steps"] == int(g["unsuccessful_steps"])
    assert abs(s["initial_cost"] - float(g["initial_cost"])) <= 1e-10 * float(g["initial_cost"])
    assert np.allclose(tr, g["cost_trace"], rtol=1e-7)
    assert abs(s["final_rms_px"] - float(g["final_rms_px"])) <= RMS_TOL_PX
    assert np.allclose(P[::10], g["poses_sample"], atol=1e-6) and np.allclose(X[::500], g["points_sample"], atol=1e-5)
    assert np.allclose(I[::10], g["intrinsics_sample"], rtol=1e-6, atol=1e-6)


def test_parameter_tolerance_is_taken_over_the_reduced_program(gpu_ctx):
    """|x| of the parameter-tolerance test counts only blocks of the reduced program (Ceres' x_norm_): shifting the
    principal point -- a constant block below 10 cameras -- together with the observations must not move the
    termination; GPU and oracle stop at the same iteration either way."""
    from reconstructor_amd import ba
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_ba import shifted_principal_point
    sc = synth_ba.make_scene(6, 120, obs_per_point=5, seed=11)
    o0, o1 = orc_ba.default_options(6), ba.default_options(gpu_ctx, 6)
    for o in (o0, o1):
        o.parameter_tolerance = 1e-7
        o.function_tolerance = 0.0
    its = []
    for shift in (0.0, 1.0e6):
        scs = shifted_principal_point(sc, shift)
        P0, I0, X0, s0 = orc_ba.solve(scs, o0)
        P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, scs, o1)
        assert s1["termination"] == s0["termination"] == 3 and s1["iterations"] == s0["iterations"]
        its.append(s1["iterations"])
    assert its[0] == its[1] >= 3


def test_edge_cases_match_oracle(gpu_ctx):
    """Two cameras (3 free camera parameters), a camera nobody observes, options away from the
    defaults: GPU follows the oracle."""
    from reconstructor_amd import ba
    # 2 cameras: camera 0 fixed, camera 1 rotation only
    sc = synth_ba.make_scene(2, 30, obs_per_point=2, seed=12)
    P0, I0, X0, s0 = orc_ba.solve(sc)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert s1["reduced_dim"] == 3 and s1["iterations"] == s0["iterations"]
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= RMS_TOL_PX
    # a camera without observations (its blocks are regularised by the LM diagonal only)
    sc = synth_ba.make_scene(12, 200, obs_per_point=5, seed=13)
    keep = sc["obs_cam"] != 7
    sc["obs_cam"], sc["obs_pt"], sc["obs_uv"] = sc["obs_cam"][keep], sc["obs_pt"][keep], np.ascontiguousarray(sc["obs_uv"][keep])
    P0, I0, X0, s0 = orc_ba.solve(sc)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert s1["iterations"] == s0["iterations"] and abs(s1["final_rms_px"] - s0["final_rms_px"]) <= RMS_TOL_PX
    assert np.allclose(P1[7], sc["poses"][7], atol=1e-9)
    # no Jacobi scaling, tiny trust region, nothing fixed but the defaults of the gauge
    o0 = orc_ba.default_options(12)
    o0.jacobi_scaling = 0
    o0.initial_trust_region_radius = 10.0
    o0.max_iterations = 8
    o1 = ba.default_options(gpu_ctx, 12)
    o1.jacobi_scaling = 0
    o1.initial_trust_region_radius = 10.0
    o1.max_iterations = 8
    sc = synth_ba.make_scene(12, 200, obs_per_point=6, seed=14)
    P0, I0, X0, s0 = orc_ba.solve(sc, o0)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc, o1)
    assert s1["iterations"] == s0["iterations"] and s1["termination"] == s0["termination"]
    assert np.allclose(s1["cost_trace"], s0["cost_trace"], rtol=1e-7)


def test_solve_is_bit_reproducible(gpu_ctx):
    """The Schur complement is built by gathers in a fixed order (no float atomics anywhere in
    the solve): two runs on the same input agree bit for bit."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(60, 6000, obs_per_point=8, seed=21)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    P2, I2, X2, s2 = ba.solve_scene(gpu_ctx, sc)
    assert np.array_equal(P1, P2) and np.array_equal(I1, I2) and np.array_equal(X1, X2)
    assert np.array_equal(s1["cost_trace"], s2["cost_trace"])


def test_pair_lists_are_kept_for_the_same_graph_only(gpu_ctx):
    """rcn_ba_solve keeps the observation-pair lists of the Schur build for the next solve of the SAME graph (arrays compared
    element by element): the same bits either way, and any change of the graph -- one observation moved to another camera,
    another scene, a session solve in between -- rebuilds them."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(30, 2500, obs_per_point=7, seed=33)
    other = synth_ba.make_scene(5, 80, obs_per_point=4, seed=34)
    ba.solve_scene(gpu_ctx, other)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    P2, I2, X2, s2 = ba.solve_scene(gpu_ctx, sc)
    assert (s1["pair_lists_reused"], s2["pair_lists_reused"]) == (0, 1)
    assert P1.tobytes() == P2.tobytes() and I1.tobytes() == I2.tobytes() and X1.tobytes() == X2.tobytes()
    assert np.array_equal(s1["cost_trace"], s2["cost_trace"])
    # another start on the same graph: still reused, and equal to a solve that builds the lists
    sc_b = dict(sc)
    sc_b["points"] = sc["points"] + 1e-3
    Pb, Ib, Xb, sb = ba.solve_scene(gpu_ctx, sc_b)
    assert sb["pair_lists_reused"] == 1
    ba.solve_scene(gpu_ctx, other)
    Pc, Ic, Xc, s_c = ba.solve_scene(gpu_ctx, sc_b)
    assert s_c["pair_lists_reused"] == 0 and Pb.tobytes() == Pc.tobytes() and Xb.tobytes() == Xc.tobytes()
    # one observation seen by another camera (the arrays stay landmark-major): not the same graph
    sc_m = dict(sc)
    cam = sc["obs_cam"].copy()
    j = sc["obs_pt"][0]
    used = set(cam[sc["obs_pt"] == j].tolist())
    cam[0] = next(c for c in range(30) if c not in used)
    sc_m["obs_cam"] = cam
    Pm, Im, Xm, sm = ba.solve_scene(gpu_ctx, sc_m)
    assert sm["pair_lists_reused"] == 0
    P0, I0, X0, s0 = orc_ba.solve(sc_m, threads=4)
    assert sm["iterations"] == s0["iterations"] and abs(sm["final_rms_px"] - s0["final_rms_px"]) <= 1e-5


def test_factorisation_kernels_one_by_one(gpu_ctx):
    """The kernels round 5 added to the factorisation, each against host arithmetic (tools/chol_kernels_check, built by
    __graft_entry__.build()): the rolled trailing update at K = 192 .. 1024, the panel product with a super-block's inverse (pipelined
    and latency form), the latency-form multi-panel update, the block rows of a super-block's inverse for g = 4 and 8."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "chol_kernels_check")
    assert os.path.exists(exe), "tools/chol_kernels_check missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    assert r.stdout.count(" ok") >= 12 and "FAILED" not in r.stdout


def test_consecutive_invalid_steps_end_the_solve_where_the_oracle_ends_it(gpu_ctx):
    """the boundary of max_consecutive_invalid_steps.  Every step of this scene is invalid (a landmark block that is exactly
    zero, no floor under the LM diagonal): the solve ends in FAILURE after exactly `limit` steps -- Ceres' `>=` -- with the parameters
    untouched, on the GPU as in the oracle; with the default floor both solve it."""
    from reconstructor_amd import ba
    sc = synth_ba.make_always_invalid_scene()
    for limit in (1, 5, 6):
        o = ba.default_options(gpu_ctx, 6)
        o.min_lm_diagonal = 0.0
        o.max_consecutive_invalid_steps = limit
        P, I, X, s = ba.solve_scene(gpu_ctx, sc, o, allow_failure=True)
        oo = orc_ba.default_options(6)
        oo.min_lm_diagonal = 0.0
        oo.max_consecutive_invalid_steps = limit
        P0, I0, X0, s0 = orc_ba.solve(sc, options=oo, threads=2)
        assert s["termination"] == 6 == s0["termination"]
        assert s["invalid_steps"] == limit == s0["invalid_steps"] and s["iterations"] == limit == s0["iterations"]
        assert s["final_cost"] == s["initial_cost"] and np.array_equal(P, sc["poses"]) and np.array_equal(X, sc["points"])
    P, I, X, s = ba.solve_scene(gpu_ctx, sc)
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=2)
    assert s["termination"] == s0["termination"] != 6 and s["iterations"] == s0["iterations"] and s["invalid_steps"] == 0
    assert abs(s["final_rms_px"] - s0["final_rms_px"]) <= 1e-5


def test_pair_lists_are_rebuilt_when_the_options_free_another_camera(gpu_ctx):
    """the pair lists also depend on the options -- a camera without free parameters (camera 0 with its pose fixed and all
    intrinsics constant) has no pairs in them.  The same graph solved again with the bounded-intrinsics branch, in which camera 0
    has four free columns, must rebuild the lists: the result equals a solve that never saw the first one, and the oracle's."""
    import ctypes as C
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(8, 300, obs_per_point=4, seed=91)
    o0 = ba.default_options(gpu_ctx, 8)
    assert o0.intrinsics_mode == 0 and o0.fix_cam0_pose == 1          # fewer than ten cameras: camera 0 has no free parameter
    P0, I0, X0, s0 = ba.solve_scene(gpu_ctx, sc, o0)
    o1 = ba.default_options(gpu_ctx, 8)
    o1.intrinsics_mode = 1                                            # same graph, camera 0 now carries fx, fy, k1, k2
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc, o1)
    assert s1["pair_lists_reused"] == 0, "lists built without camera 0's pairs must not serve a solve in which it is free"
    other = synth_ba.make_scene(5, 80, obs_per_point=4, seed=34)
    ba.solve_scene(gpu_ctx, other)                                    # drops the lists: the next solve builds them from scratch
    P2, I2, X2, s2 = ba.solve_scene(gpu_ctx, sc, o1)
    assert s2["pair_lists_reused"] == 0
    assert P1.tobytes() == P2.tobytes() and I1.tobytes() == I2.tobytes() and X1.tobytes() == X2.tobytes()
    oo = orc_ba.default_options(8)
    oo.intrinsics_mode = 1
    Pr, Ir, Xr, sr = orc_ba.solve(sc, options=oo, threads=4)
    assert s1["iterations"] == sr["iterations"] and abs(s1["final_rms_px"] - sr["final_rms_px"]) <= 1e-5
    # and back: the first options again, lists rebuilt once more, same bits as the very first solve
    P3, I3, X3, s3 = ba.solve_scene(gpu_ctx, sc, o0)
    assert s3["pair_lists_reused"] == 0 and P3.tobytes() == P0.tobytes() and X3.tobytes() == X0.tobytes()
    P4, I4, X4, s4 = ba.solve_scene(gpu_ctx, sc, o0)
    assert s4["pair_lists_reused"] == 1 and P4.tobytes() == P0.tobytes()


_ALT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch
from reconstructor_amd import _lib, ba, synth_ba
assert b"DIAGNOSTIC" in _lib.load().rcn_version()
sc = synth_ba.make_scene(40, 3000, obs_per_point=6, seed=33)
P, I, X, s = ba.solve_scene(_lib.Context(0), sc)
np.savez(sys.argv[1], P=P, X=X, it=s["iterations"], rms=s["final_rms_px"])
"""


@pytest.mark.parametrize("env", ["RCN_BA_TRSV_FWD", "RCN_BA_SCHUR_ATOMICS"])
def test_alternative_device_paths_agree(gpu_ctx, env, tmp_path):
    """The default solve carries the right-hand side through the factorisation as an extra row and
    builds the Schur complement by MFMA gathers; the separate forward substitution and the atomic
    Schur build must land on the same optimum.  Those alternatives exist only in the diagnostic build
    (tools/librcn_diag.so, -DRCN_DIAG), whose rcn_create reads the switches: a child process loads it."""
    import os, subprocess, sys
    from reconstructor_amd import ba
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "tools", "librcn_diag.so")
    if not os.path.exists(diag):
        pytest.skip("tools/librcn_diag.so not built")
    sc = synth_ba.make_scene(40, 3000, obs_per_point=6, seed=33)
    P0, I0, X0, s0 = ba.solve_scene(gpu_ctx, sc)
    out = str(tmp_path / "alt.npz")
    r = subprocess.run([sys.executable, "-c", _ALT % root, out], env=dict(os.environ, RCN_LIB=diag, **{env: "1"}),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    alt = np.load(out)
    assert int(alt["it"]) == s0["iterations"]
    assert abs(float(alt["rms"]) - s0["final_rms_px"]) < 1e-9
    assert np.allclose(alt["P"], P0, rtol=0, atol=1e-8) and np.allclose(alt["X"], X0, rtol=0, atol=1e-7)


def test_rejected_steps_follow_the_oracle(gpu_ctx):
    """A badly initialised, noisy scene: the trust region shrinks through unsuccessful steps; the GPU loop
    must take the same accept / reject decisions as the CPU restatement (same iteration trace)."""
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(10, 200, obs_per_point=5, seed=40, noise_px=2.0, perturb=(0.8, 3.0, 2.5))   # 47 accepted, 3 rejected, hits the 50-iteration cap
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=4)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert s0["unsuccessful_steps"] + s0["invalid_steps"] > 0, "the scene is meant to produce rejected steps"
    assert s1["iterations"] == s0["iterations"] and s1["successful_steps"] == s0["successful_steps"]
    assert s1["unsuccessful_steps"] == s0["unsuccessful_steps"] and s1["termination"] == s0["termination"]
    n = min(len(s0["cost_trace"]), len(s1["cost_trace"]), 12)
    assert np.allclose(s1["cost_trace"][:n], s0["cost_trace"][:n], rtol=1e-6)       # the early trace; 50 chaotic iterations amplify rounding
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= 1e-3 * s0["final_rms_px"]


def test_far_start_at_cfg4_size_follows_the_oracle(gpu_ctx):
    """The bench's far-start variant (ba_bench.FAR_START) at cfg-4 size, where the oracle finishes in seconds: 50 iterations
    into the cap, the bounds line search on most of them, a rejected step on the way.  Same accept / reject decisions and
    the same early cost trace; 50 badly conditioned iterations amplify rounding, so the end is compared loosely."""
    from reconstructor_amd import ba, ba_bench
    sc = synth_ba.make_scene(200, 20000, obs_per_point=10, seed=2024, perturb=ba_bench.FAR_START)
    P0, I0, X0, s0 = orc_ba.solve(sc, threads=6)
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert s0["iterations"] >= 10 and s0["line_search_backtracks"] > 0, "the scene is meant to be a long, line-searched solve"
    assert s1["iterations"] == s0["iterations"] and s1["termination"] == s0["termination"]
    assert s1["successful_steps"] == s0["successful_steps"] and s1["unsuccessful_steps"] == s0["unsuccessful_steps"]
    assert np.allclose(s1["cost_trace"][:3], s0["cost_trace"][:3], rtol=1e-6)       # before the first interpolated line-search step
    n = min(len(s0["cost_trace"]), len(s1["cost_trace"]), 12)
    assert np.allclose(s1["cost_trace"][:n], s0["cost_trace"][:n], rtol=1e-2)       # interpolated steps amplify rounding (1e-4 by the fourth entry)
    assert abs(s1["final_rms_px"] - s0["final_rms_px"]) <= 1e-2 * s0["final_rms_px"]


def test_a_broken_stream_hand_off_falls_back_to_one_stream_with_the_same_bits(gpu_ctx):
    """a cross-stream wait of the factorisation that gives up must cost time, not the result.  The diagnostic
    build can break one hand-off on purpose (RCN_CHOL_BREAK=1: step 1 of every three-stream factorisation waits for a count
    that never comes); the wait times out after 2 s, the context latches the one-stream schedule, the linear solve of that
    iteration is redone -- and poses, intrinsics and points equal the shipping library's, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from reconstructor_amd import ba
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "tools/librcn_diag.so missing: run __graft_entry__.build()"
    sc = synth_ba.make_scene(70, 1500, obs_per_point=6, seed=77)         # 6 blocks of 128: every kind of step occurs
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert s1["factor_schedule"] == 0
    code = ("import sys, json, hashlib; sys.path.insert(0, %r)\n"
            "from reconstructor_amd import _lib, ba, synth_ba\n"
            "sc = synth_ba.make_scene(70, 1500, obs_per_point=6, seed=77)\n"
            "P, I, X, s = ba.solve_scene(_lib.Context(0), sc)\n"
            "print(json.dumps({'h': hashlib.sha256(P.tobytes() + I.tobytes() + X.tobytes()).hexdigest(), 'sched': s['factor_schedule'],\n"
            "                  'it': s['iterations'], 'cost': s['final_cost'], 'sec': s['solve_seconds'], 'invalid': s['invalid_steps']}))\n" % root)
    env = dict(os.environ, RCN_LIB=diag, RCN_CHOL_BREAK="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    import hashlib
    assert d["sched"] == 1 and d["sec"] > 1.5 and d["invalid"] == 0
    assert d["it"] == s1["iterations"] and d["cost"] == s1["final_cost"]
    assert d["h"] == hashlib.sha256(P1.tobytes() + I1.tobytes() + X1.tobytes()).hexdigest()
    # the kernels behind a timed-out hand-off run on half-updated tiles and may meet a non-finite pivot.  The
    # timeout must stay in the flag word (RCN_CHOL_BREAK=2 puts a NaN on the next diagonal block behind the broken wait): the
    # context still switches schedules instead of counting a failed step, and every later wait of that factorisation returns
    # at once -- ONE 2-s timeout in the whole solve, not one per hand-off.
    env = dict(os.environ, RCN_LIB=diag, RCN_CHOL_BREAK="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["sched"] == 1 and d["invalid"] == 0 and 1.5 < d["sec"] < 4.0, d
    assert d["it"] == s1["iterations"] and d["h"] == hashlib.sha256(P1.tobytes() + I1.tobytes() + X1.tobytes()).hexdigest()
    # the backward substitution as one launch (x handed from block row to block row through the data itself) against the
    # per-step kernels it falls back to: the same bits
    env = dict(os.environ, RCN_LIB=diag, RCN_TRSV_CHAIN="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["sched"] == 0 and d["invalid"] == 0 and d["h"] == hashlib.sha256(P1.tobytes() + I1.tobytes() + X1.tobytes()).hexdigest()


_SMALL = """
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import torch
from reconstructor_amd import _lib, ba, synth_ba
ctx = _lib.Context(0)
h = hashlib.sha256()
for nc, npts, k in ((3, 40, 3), (6, 300, 5), (9, 150, 6), (12, 700, 8), (25, 1500, 8)):
    sc = synth_ba.make_scene(nc, npts, obs_per_point=k, seed=61)
    P, I, X, s = ba.solve_scene(ctx, sc)
    h.update(P.tobytes() + I.tobytes() + X.tobytes() + repr((s["iterations"], s["termination"], s["final_cost"])).encode())
print(h.hexdigest())
"""


def test_small_solve_shortcuts_change_no_bit(gpu_ctx):
    """Round 5: the LM step's scalars reach the host through a pinned mirror written by the step's last kernel, and the smallest graphs'
    pair lists are counted, scanned and filled by one workgroup.  Both are shortcuts around the same arithmetic: with either switched
    off (diagnostic build, child process) five small solves give the product's bits."""
    import hashlib, os, subprocess, sys
    from reconstructor_amd import ba
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "tools/librcn_diag.so missing: run __graft_entry__.build()"
    h = hashlib.sha256()
    for nc, npts, k in ((3, 40, 3), (6, 300, 5), (9, 150, 6), (12, 700, 8), (25, 1500, 8)):
        sc = synth_ba.make_scene(nc, npts, obs_per_point=k, seed=61)
        P, I, X, s = ba.solve_scene(gpu_ctx, sc)
        h.update(P.tobytes() + I.tobytes() + X.tobytes() + repr((s["iterations"], s["termination"], s["final_cost"])).encode())
    for env in ({"RCN_BA_MIRROR": "0"}, {"RCN_PAIR_SMALL": "0"}):
        r = subprocess.run([sys.executable, "-c", _SMALL % root], env=dict(os.environ, RCN_LIB=diag, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.strip().splitlines()[-1] == h.hexdigest(), env


@pytest.mark.parametrize("env", [{"RCN_CHOL_DIAG_SERVER": "1"}, {"RCN_CHOL_BULK_BEHIND": "1"}, {"RCN_CHOL_GATE_IN_KERNEL": "1"},
                                 {"RCN_CHOL_GATE_IN_KERNEL": "-1"}, {"RCN_CHOL_DIAG_SERVER": "1", "RCN_POLL_MODE": "1"}])
def test_where_and_when_an_operation_runs_changes_no_bit(gpu_ctx, env):
    """The factorisation's schedule is data (csrc/chol_plan.h), and no operation's arithmetic depends on where or when it runs: the plan
    options that only move operations between streams or reorder the list -- the diagonal blocks in one resident workgroup, bulk updates
    behind the next diagonal block, waits inside the kernels or in gate kernels in front -- give the shipping schedule's bits
    (diagnostic build, child process).  (Options that hand tiles to ANOTHER kernel -- the two-level regime's thresholds, the carved
    tiles -- give their own, equally repeatable bits: tools/soak_ba_large.py solves every problem twice under them.)"""
    import hashlib, json, os, subprocess, sys
    from reconstructor_amd import ba
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "tools/librcn_diag.so missing: run __graft_entry__.build()"
    sc = synth_ba.make_scene(260, 5200, obs_per_point=6, seed=78)         # 18 blocks of 128
    P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, sc)
    assert s1["factor_schedule"] == 0 and s1["invalid_steps"] == 0
    code = ("import sys, json, hashlib; sys.path.insert(0, %r)\n"
            "from reconstructor_amd import _lib, ba, synth_ba\n"
            "sc = synth_ba.make_scene(260, 5200, obs_per_point=6, seed=78)\n"
            "P, I, X, s = ba.solve_scene(_lib.Context(0), sc)\n"
            "print(json.dumps({'h': hashlib.sha256(P.tobytes() + I.tobytes() + X.tobytes()).hexdigest(), 'sched': s['factor_schedule'], 'it': s['iterations']}))\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, RCN_LIB=diag, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["sched"] == 0 and d["it"] == s1["iterations"]
    assert d["h"] == hashlib.sha256(P1.tobytes() + I1.tobytes() + X1.tobytes()).hexdigest()

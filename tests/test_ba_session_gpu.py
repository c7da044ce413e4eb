"""GPU suite: device-resident incremental bundle adjustment (rcn_ba_session_*, csrc/ba_session.hip): the reference's loop --
a global BA after every registered view, SequentialReconstructor.cpp:1040-1094 -- with the problem kept in HBM between
the solves.  Every session solve is compared, bit for bit, with rcn_ba_solve on the same problem packed from scratch,
and with the CPU oracle to the north_star tolerance; the validity sweep / outlier removal on the session's arrays with
the oracle's sweep."""
import numpy as np
import pytest

from oracle import orc_ba, orc_validity
from reconstructor_amd import synth_ba

pytestmark = pytest.mark.gpu


def _view_additions(sc, n, live_before):
    """What registering camera n-1 adds, as the reference's addNextView would deliver it: landmarks that now have >= 2
    observations among cameras 0..n-1 (with all of those observations) and, for landmarks already there, the new camera's own."""
    keep = sc["obs_cam"] < n
    cnt = np.bincount(sc["obs_pt"][keep], minlength=len(sc["points"]))
    live = cnt >= 2
    new_pts = np.flatnonzero(live & ~live_before)
    return live, new_pts


def test_incremental_loop_through_a_session(gpu_ctx):
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(25, 1500, obs_per_point=6, seed=31)
    xy_all = sc["obs_uv"].astype(np.int32)
    ses = ba.BaSession(gpu_ctx)
    try:
        sid = {}                                        # scene landmark -> session landmark
        live = np.zeros(len(sc["points"]), bool)
        reused = []
        for n in range(1, 26):
            ses.add_camera(sc["poses"][n - 1], sc["intrinsics"][n - 1])
            live_now, new_pts = _view_additions(sc, n, live)
            # observations of the new camera on landmarks that were already there: appended to their tracks
            o_new = np.flatnonzero((sc["obs_cam"] == n - 1) & live[sc["obs_pt"]])
            ses.add_observations([sid[j] for j in sc["obs_pt"][o_new]], sc["obs_cam"][o_new], xy_all[o_new])
            if len(new_pts):
                first = ses.add_points(sc["points"][new_pts])
                for k, j in enumerate(new_pts):
                    sid[j] = first + k
                o_tr = np.flatnonzero(np.isin(sc["obs_pt"], new_pts) & (sc["obs_cam"] < n))
                ses.add_observations([sid[j] for j in sc["obs_pt"][o_tr]], sc["obs_cam"][o_tr], xy_all[o_tr])
            live = live_now
            if n < 3:
                continue
            # ---- the same problem, packed from scratch, through rcn_ba_solve and through the oracle
            poses, intr = ses.cameras()
            pt, cam, xy = ses.graph()
            X = ses.points()
            nc, npts, no = ses.counts()
            assert (nc, npts, no) == (n, len(sid), len(pt)) and (np.diff(pt) >= 0).all()
            flat = {"poses": poses, "intrinsics": intr, "points": X, "obs_uv": xy.astype(np.float64), "obs_cam": cam, "obs_pt": pt}
            P1, I1, X1, s1 = ba.solve_scene(gpu_ctx, flat)
            P0, I0, X0, s0 = orc_ba.solve(flat, threads=4)
            s2 = ses.solve()                                      # graph changed since the last solve: lists rebuilt
            P2, I2 = ses.cameras()
            X2 = ses.points()
            assert s2["pair_lists_reused"] == 0
            assert P2.tobytes() == P1.tobytes() and I2.tobytes() == I1.tobytes() and X2.tobytes() == X1.tobytes()
            assert s2["iterations"] == s1["iterations"] and np.array_equal(s2["cost_trace"], s1["cost_trace"])
            assert s2["iterations"] == s0["iterations"] and s2["termination"] == s0["termination"]
            assert abs(s2["final_rms_px"] - s0["final_rms_px"]) <= 1e-5
            assert s2["reduced_dim"] == (6 * (n - 1) - 3 if n < 10 else 6 * (n - 1) - 3 + 4 * n)
            if n in (5, 12):                                      # a second solve on the unchanged graph reuses the pair lists
                s3 = ses.solve()
                reused.append(s3["pair_lists_reused"])
                assert s3["iterations"] >= 0 and s3["final_cost"] <= s2["final_cost"] * (1 + 1e-12)
        assert reused == [1, 1]
    finally:
        ses.close()


def test_validity_sweep_and_outlier_removal_on_the_session_arrays(gpu_ctx):
    from reconstructor_amd import ba
    c = synth_ba.make_validity_case(14, 900, obs_per_point=6, seed=7, defect_rate=0.15)
    ses = ba.BaSession(gpu_ctx)
    try:
        # poses34 are given to the sweep as the pipeline holds them; camera parameters of the session only feed the solver
        for l in range(14):
            ses.add_camera(np.zeros(6), c["intrinsics"][l])
        ses.add_points(c["points"])
        pt = np.repeat(np.arange(900), np.diff(c["pt_off"])).astype(np.int32)
        ses.add_observations(pt, c["obs_cam"], c["obs_xy"])
        inl0, keep0 = orc_validity.landmark_validity(**c)
        inl, erased = ses.validity(c["poses34"], 4.0, 1.0)
        assert np.array_equal(inl, inl0) and erased == int((~keep0).sum()) and 0 < inl.sum() < 900
        # the erased observations are gone from the tracks, in order
        pt2, cam2, xy2 = ses.graph()
        assert np.array_equal(pt2, pt[keep0]) and np.array_equal(cam2, c["obs_cam"][keep0]) and np.array_equal(xy2, c["obs_xy"][keep0])
        new_idx, removed = ses.remove_outliers()
        assert removed == int((~inl0).sum()) and np.array_equal(new_idx >= 0, inl0)
        assert np.array_equal(new_idx[inl0], np.arange(inl0.sum()))
        assert ses.points().tobytes() == np.ascontiguousarray(c["points"][inl0]).tobytes()      # compacted on the device
        pt3, cam3, xy3 = ses.graph()
        sel = keep0 & inl0[pt]
        assert np.array_equal(pt3, new_idx[pt[sel]]) and np.array_equal(cam3, c["obs_cam"][sel])
        # a second sweep on the cleaned graph: the reference's erase loop skips the element that slides into an erased slot,
        # so it is NOT idempotent -- the session must again do what the oracle does on the same (cleaned) arrays
        c2 = dict(c, points=c["points"][inl0], obs_cam=cam3, obs_xy=xy3,
                  pt_off=np.concatenate([[0], np.cumsum(np.bincount(pt3, minlength=int(inl0.sum())))]).astype(np.int32))
        inl0b, keep0b = orc_validity.landmark_validity(**c2)
        inl_b, erased_b = ses.validity(c["poses34"], 4.0, 1.0)
        assert np.array_equal(inl_b, inl0b) and erased_b == int((~keep0b).sum())
        with pytest.raises(Exception):
            ses.add_observations([10 ** 6], [0], [[1, 1]])
    finally:
        ses.close()

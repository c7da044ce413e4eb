"""CPU suite: the landmark-validity oracle (oracle/validity_oracle.c) against hand-made known
answers, the committed golden fixture and an independent list-based transcription."""
import os

import numpy as np
import pytest

import indep
from oracle import orc_validity as ov
from reconstructor_amd import synth_ba, validity

GOLD = os.path.join(os.path.dirname(__file__), "golden", "validity_small.npz")
I34 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], float)          # camera at the origin looking down +z
K0 = np.array([600.0, 600.0, 256.0, 168.0, 0.0, 0.0])


def cam_at(x):   # same orientation, centre (x, 0, 0):  t = -C
    p = I34.copy(); p[3] = -x
    return p


def test_projection_error_and_angle_known_answers():
    e, depth = ov.projection_error(I34, K0, [0.5, 0.2, 5.0], 316, 192)     # u = 60 + 256, v = 24 + 168
    assert e == 0.0 and depth == 5.0
    e, _ = ov.projection_error(I34, K0, [0.5, 0.2, 5.0], 314, 195)
    assert e == 5.0
    # distortion is ADDED to x and y (Camera.h:66-69): k1 = 0.1, r = 0.0116 -> +0.00116
    e, _ = ov.projection_error(I34, np.array([600, 600, 256, 168, 0.1, 0.0]), [0.5, 0.2, 5.0], 316, 192)
    assert abs(e - 2 * 600 * 0.1 * 0.0116) < 1e-9
    # two cameras one unit apart, landmark 5 ahead on the bisector: angle = 2 atan(0.1), "degrees" with 3.1415
    a = ov.triangulation_angle(cam_at(-0.5), cam_at(0.5), [0.0, 0.0, 5.0])
    assert abs(a - 180.0 * 2 * np.arctan(0.1) / 3.1415) < 1e-12
    assert np.isnan(ov.triangulation_angle(I34, I34, [0.0, 0.0, 0.0]))      # landmark on the centre: 0/0


def _track(cams, xy, X=(0.0, 0.0, 5.0)):
    P = np.stack(cams)
    return dict(poses34=P, intrinsics=np.tile(K0, (len(P), 1)), points=np.array([X]), pt_off=[0, len(cams)],
                obs_cam=np.arange(len(cams), dtype=np.int32), obs_xy=np.array(xy, np.int32))


def test_erase_skips_the_element_that_slides_in():
    """Observations 1 and 2 are both 100 px off.  Erasing #1 moves #2 into slot 1 and the loop steps
    to slot 2, so #2 is never examined and stays (SequentialReconstructor.cpp:877-898)."""
    cams = [cam_at(x) for x in (-1.0, -0.5, 0.0, 0.5)]
    good = [256 + int(600 * (0.0 - c) / 5.0) for c in (-1.0, -0.5, 0.0, 0.5)]
    xy = [[good[0], 168], [good[1] + 100, 168], [good[2] + 100, 168], [good[3], 168]]
    inl, keep = ov.landmark_validity(**_track(cams, xy))
    assert keep.tolist() == [True, False, True, True] and inl.tolist() == [True]


def test_fewer_than_two_after_an_erase_and_angle_rules():
    cams = [cam_at(-0.5), cam_at(0.5)]
    ok = [[256 + 60, 168], [256 - 60, 168]]
    assert ov.landmark_validity(**_track(cams, ok))[0].tolist() == [True]
    bad = [[256 + 60, 168], [256 - 60 + 9, 168]]                          # second one 9 px off
    inl, keep = ov.landmark_validity(**_track(cams, bad))
    assert inl.tolist() == [False] and keep.tolist() == [True, False]
    # exactly at the threshold: 4.0 is not > 4.0
    edge = [[256 + 60 + 4, 168], [256 - 60, 168]]
    assert ov.landmark_validity(**_track(cams, edge))[0].tolist() == [True]
    # narrow baseline: 0.01 apart at depth 5 -> 0.11 "degrees" < 1
    near = [cam_at(-0.005), cam_at(0.005)]
    inl, keep = ov.landmark_validity(**_track(near, [[256, 168], [256, 168]]))
    assert inl.tolist() == [False] and keep.tolist() == [True, True]
    # behind the camera: depth < 0 erases even with zero reprojection error
    inl, keep = ov.landmark_validity(**_track(cams, [[256 + 60, 168], [256 - 60, 168]], X=(0.0, 0.0, -5.0)))
    assert keep.tolist() == [False, True] and inl.tolist() == [False]      # the erase-skip leaves #1
    # depth exactly 0: NaN/inf comparisons are false, the observation stays
    inl, keep = ov.landmark_validity(**_track(cams, ok, X=(0.0, 0.0, 0.0)))
    assert keep.tolist() == [True, True]
    # empty and single-observation tracks are outliers (no pair passes the angle test)
    one = _track([I34], [[256, 168]])
    assert ov.landmark_validity(**one)[0].tolist() == [False]
    none = dict(one, pt_off=[0, 0])
    assert ov.landmark_validity(**none)[0].tolist() == [False]


def test_golden_fixture():
    g = np.load(GOLD)
    inl, keep = ov.landmark_validity(g["poses34"], g["intrinsics"], g["points"], g["pt_off"], g["obs_cam"], g["obs_xy"])
    assert (inl == g["inlier"]).all() and (keep == g["keep"]).all()
    assert 0.5 < inl.mean() < 0.95 and (~keep).sum() > 100            # the fixture exercises both outcomes


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_equals_list_based_transcription(seed):
    c = synth_ba.make_validity_case(10, 250, obs_per_point=6, seed=seed, defect_rate=0.3)
    inl, keep = ov.landmark_validity(**c)
    inl2, keep2 = indep.validity_python(c["poses34"], c["intrinsics"], c["points"], c["pt_off"], c["obs_cam"], c["obs_xy"])
    assert (inl == inl2).all() and (keep == keep2).all()


def test_thresholds_are_parameters():
    c = synth_ba.make_validity_case(10, 300, obs_per_point=6, seed=5)
    strict = ov.landmark_validity(**c, max_err=1.0, min_angle=5.0)
    loose = ov.landmark_validity(**c, max_err=50.0, min_angle=0.01)
    assert strict[0].sum() < loose[0].sum() and strict[1].sum() < loose[1].sum()


def test_remove_outlier_landmarks_compacts_in_order():
    g = np.load(GOLD)
    pts, off, cam, xy, old = validity.remove_outlier_landmarks(g["points"], g["pt_off"], g["obs_cam"], g["obs_xy"], g["inlier"], g["keep"])
    assert len(pts) == g["inlier"].sum() == len(off) - 1 and off[-1] == len(cam) == len(xy)
    j = int(old[7])
    o = np.arange(g["pt_off"][j], g["pt_off"][j + 1])[g["keep"][g["pt_off"][j]:g["pt_off"][j + 1]]]
    assert (cam[off[7]:off[8]] == g["obs_cam"][o]).all() and (pts[7] == g["points"][j]).all()
    # a second sweep over the cleaned graph erases nothing more if nothing was skipped... it may
    # (the erase-skip quirk), but every landmark it keeps was an inlier of the first sweep
    inl2, _ = ov.landmark_validity(g["poses34"], g["intrinsics"], pts, off, cam, xy)
    assert inl2.sum() <= len(pts)

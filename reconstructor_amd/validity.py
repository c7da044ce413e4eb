"""Host mirror of the reference's landmark validity sweep over librcn.so (no CPU fallback).

    SequentialReconstructor::checkLandmarkValidity     SequentialReconstructor.cpp:869-954
    SequentialReconstructor::removeOutlierLandmarks    SequentialReconstructor.cpp:956-976
"""
import ctypes as C

import numpy as np

from . import _lib

MAX_PROJECTION_ERROR = 4.0      # SequentialReconstructor.h:256
MIN_TRIANGULATION_ANGLE = 1.0   # SequentialReconstructor.h:257


def check_landmark_validity(ctx, poses34, intrinsics, points, pt_off, obs_cam, obs_xy,
                            max_projection_error=MAX_PROJECTION_ERROR,
                            min_triangulation_angle=MIN_TRIANGULATION_ANGLE):
    """Flat form of checkLandmarkValidity: returns (inlier[n_points] bool, keep[n_obs] bool).

    `keep` is what the reference does to landmark.triangulatedFeatures in place (erased
    observations are False); `inlier` is the vector<bool> it returns."""
    poses34 = np.ascontiguousarray(poses34, np.float64).reshape(-1, 12)
    intrinsics = np.ascontiguousarray(intrinsics, np.float64).reshape(-1, 6)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    pt_off = np.ascontiguousarray(pt_off, np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, np.int32)
    obs_xy = np.ascontiguousarray(obs_xy, np.int32).reshape(-1, 2)
    if len(poses34) != len(intrinsics):
        raise ValueError("one intrinsics row per pose")
    if len(pt_off) != len(points) + 1 or len(obs_xy) != len(obs_cam):
        raise ValueError("pt_off needs n_points + 1 entries; obs_xy one row per observation")
    pb = _lib.LandmarkProblem(len(poses34), len(points), len(obs_cam), 0,
                              poses34.ctypes.data, intrinsics.ctypes.data, points.ctypes.data,
                              pt_off.ctypes.data, obs_cam.ctypes.data, obs_xy.ctypes.data)
    inl = np.zeros(len(points), np.uint8)
    keep = np.zeros(len(obs_cam), np.uint8)
    n = C.c_int32(0)
    ctx.check(ctx.lib.rcn_landmark_validity(ctx.h, C.byref(pb), float(max_projection_error),
                                            float(min_triangulation_angle), inl.ctypes.data,
                                            keep.ctypes.data, C.addressof(n)))
    return inl.astype(bool), keep.astype(bool)


def remove_outlier_landmarks(points, pt_off, obs_cam, obs_xy, inlier, keep):
    """removeOutlierLandmarks on the flat arrays: inlier landmarks with their surviving
    observations, in order.  Returns (points, pt_off, obs_cam, obs_xy, old_index_of_new_landmark)."""
    points = np.asarray(points, np.float64).reshape(-1, 3)
    pt_off = np.asarray(pt_off, np.int64)
    inlier = np.asarray(inlier, bool)
    keep = np.asarray(keep, bool)
    owner = np.repeat(np.arange(len(points)), np.diff(pt_off))
    sel = keep[pt_off[0]:pt_off[-1]] & inlier[owner]
    idx = np.flatnonzero(sel) + pt_off[0]
    counts = np.bincount(owner[sel], minlength=len(points))[inlier]
    new_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return (points[inlier], new_off, np.asarray(obs_cam, np.int32)[idx],
            np.asarray(obs_xy, np.int32).reshape(-1, 2)[idx], np.flatnonzero(inlier))

"""ctypes binding of the landmark-validity oracle (oracle/validity_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

MAX_PROJECTION_ERROR = 4.0      # SequentialReconstructor.h:256
MIN_TRIANGULATION_ANGLE = 1.0   # SequentialReconstructor.h:257


def register(L):
    L.orc_landmark_validity.restype = C.c_int
    L.orc_landmark_validity.argtypes = [C.c_int, _f64p, _f64p, C.c_int, _f64p, _i32p, _i32p, _i32p,
                                        C.c_double, C.c_double, _u8p, _u8p]
    L.orc_projection_error.restype = C.c_double
    L.orc_projection_error.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.orc_triangulation_angle.restype = C.c_double
    L.orc_triangulation_angle.argtypes = [_f64p, _f64p, _f64p]


def _lib():
    from . import orc
    return orc.lib()


def landmark_validity(poses34, intrinsics, points, pt_off, obs_cam, obs_xy,
                      max_err=MAX_PROJECTION_ERROR, min_angle=MIN_TRIANGULATION_ANGLE):
    """checkLandmarkValidity (SequentialReconstructor.cpp:869-954): (inlier[n_points], keep[n_obs])."""
    poses34 = np.ascontiguousarray(poses34, np.float64).reshape(-1, 12)
    intrinsics = np.ascontiguousarray(intrinsics, np.float64).reshape(-1, 6)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    pt_off = np.ascontiguousarray(pt_off, np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, np.int32)
    obs_xy = np.ascontiguousarray(obs_xy, np.int32).reshape(-1, 2)
    inl = np.zeros(max(len(points), 1), np.uint8)
    keep = np.zeros(max(len(obs_cam), 1), np.uint8)
    n = _lib().orc_landmark_validity(len(poses34), poses34, intrinsics, len(points), points, pt_off,
                                     obs_cam if len(obs_cam) else np.zeros(1, np.int32),
                                     obs_xy if len(obs_xy) else np.zeros((1, 2), np.int32),
                                     float(max_err), float(min_angle), inl, keep)
    if n < 0:
        raise ValueError("track longer than the oracle's limit")
    return inl[:len(points)].astype(bool), keep[:len(obs_cam)].astype(bool)


def projection_error(pose34, intr, X, fx, fy):
    d = C.c_double()
    e = _lib().orc_projection_error(np.ascontiguousarray(pose34, np.float64).reshape(12), np.ascontiguousarray(intr, np.float64),
                                    np.ascontiguousarray(X, np.float64), int(fx), int(fy), C.byref(d))
    return e, d.value


def triangulation_angle(pose_a, pose_b, X):
    return _lib().orc_triangulation_angle(np.ascontiguousarray(pose_a, np.float64).reshape(12),
                                          np.ascontiguousarray(pose_b, np.float64).reshape(12),
                                          np.ascontiguousarray(X, np.float64))

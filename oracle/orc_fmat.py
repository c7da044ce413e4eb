"""ctypes binding of the epipolar-filter oracle (oracle/fmat_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def register(L):
    L.orc_fmat_filter.restype = C.c_int
    L.orc_fmat_filter.argtypes = [_i32p, _i32p, C.c_int, _u8p, C.POINTER(C.c_int32)]
    L.orc_fmat_filter_grid.restype = None
    L.orc_fmat_filter_grid.argtypes = [C.c_int, _i32p, _i32p, _i32p, _u8p, _i32p, _i32p, C.c_int]
    L.orc_fmat_7point.restype = C.c_int
    L.orc_fmat_7point.argtypes = [_f32p, _f32p, _f64p]
    L.orc_fmat_error.restype = C.c_float
    L.orc_fmat_error.argtypes = [_f64p, _f32p, _f32p]
    L.orc_rng_next.restype = C.c_uint
    L.orc_rng_next.argtypes = [C.POINTER(C.c_uint64)]
    L.orc_ransac_num_iters.restype = C.c_int
    L.orc_ransac_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]


def _lib():
    from . import orc
    return orc.lib()


def filter_pair(xy1, xy2):
    """estimateFundamental's inlier mask for one pair: (mask[n] bool, count, iterations).
    count = -1: no model (the reference drops the pair's matches); -2: fewer than 7 points (not filtered)."""
    xy1 = np.ascontiguousarray(xy1, np.int32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.int32).reshape(-1, 2)
    n = len(xy1)
    mask = np.zeros(max(n, 1), np.uint8)
    it = C.c_int32(0)
    cnt = _lib().orc_fmat_filter(xy1 if n else np.zeros((1, 2), np.int32), xy2 if n else np.zeros((1, 2), np.int32), n, mask, C.byref(it))
    return mask[:n].astype(bool), cnt, it.value


def filter_grid(off, xy1, xy2, threads=1):
    off = np.ascontiguousarray(off, np.int32)
    xy1 = np.ascontiguousarray(xy1, np.int32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.int32).reshape(-1, 2)
    P = len(off) - 1
    mask = np.zeros(max(len(xy1), 1), np.uint8)
    counts = np.zeros(max(P, 1), np.int32)
    iters = np.zeros(max(P, 1), np.int32)
    _lib().orc_fmat_filter_grid(P, off, xy1 if len(xy1) else np.zeros((1, 2), np.int32), xy2 if len(xy2) else np.zeros((1, 2), np.int32),
                                mask, counts, iters, int(threads))
    return mask[:len(xy1)].astype(bool), counts[:P], iters[:P]


def seven_point(m1, m2):
    F = np.zeros((3, 9))
    n = _lib().orc_fmat_7point(np.ascontiguousarray(m1, np.float32).reshape(7, 2), np.ascontiguousarray(m2, np.float32).reshape(7, 2), F)
    return F[:max(n, 0)].reshape(-1, 3, 3)


def epi_error(F, p1, p2):
    return _lib().orc_fmat_error(np.ascontiguousarray(F, np.float64).reshape(9), np.ascontiguousarray(p1, np.float32), np.ascontiguousarray(p2, np.float32))


def rng_sequence(n, state=(1 << 64) - 1):
    s = C.c_uint64(state)
    return [int(_lib().orc_rng_next(C.byref(s))) for _ in range(n)]


def num_iters(p, ep, model_points=7, max_iters=1000):
    return _lib().orc_ransac_num_iters(float(p), float(ep), int(model_points), int(max_iters))

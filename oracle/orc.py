"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False, native=False):
    """liborc.so: -march=x86-64-v3 (travels from the build container to the GPU box).  native=True builds
    liborc_native.so with -march=native ON THE BOX IT RUNS ON and makes it the library every later call
    uses -- bench.py's cpu_baseline legs do so (the CPU baseline is compiled -march=native);
    the arithmetic is the same either way (-ffp-contract=off, explicit fma() calls)."""
    global _LIB
    name = "liborc_native.so" if native else "liborc.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("match_oracle.c", "ba_oracle.c", "validity_oracle.c", "fmat_oracle.c", "desc_oracle.c", "Makefile")] + \
           [os.path.join(os.path.dirname(_HERE), "include", "rcn.h")]      # ba_oracle.c shares the option / summary struct layouts
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if native and getattr(build, "_native_done", False):
        return so
    if force or stale or native:          # a native build is never trusted from another machine
        subprocess.check_call(["make", "-B", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    if native:
        build._native_done = True
        _LIB = C.CDLL(so)
        _sig()
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _sig()
    return _LIB


_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def _sig():
    L = _LIB
    L.orc_l2sq.restype = C.c_double
    L.orc_l2sq.argtypes = [_f32p, _f32p, C.c_int]
    L.orc_knn2.restype = None
    L.orc_knn2.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, _i32p, _f64p]
    L.orc_match_pair.restype = C.c_int
    L.orc_match_pair.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_float, _i32p]
    L.orc_match_grid.restype = None
    L.orc_match_grid.argtypes = [_f32p, _i64p, C.c_int, C.c_int, _i32p, C.c_int, C.c_float,
                                 _i32p, C.c_int64, _i32p, C.c_int]
    L.orc_desc_sample.restype = None
    L.orc_desc_sample.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int64, _i32p, C.c_int, C.c_int, _f32p]
    from . import orc_ba, orc_fmat, orc_validity
    orc_ba.register(L)
    orc_validity.register(L)
    orc_fmat.register(L)


RATIO = np.float32(0.7)  # FeatureMatcher.h:45  `const float ratioThresh = 0.7`


def l2sq(q, t):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    return lib().orc_l2sq(q, t, q.shape[0])


def knn2(q, t):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    K1, D = q.shape
    K2 = t.shape[0]
    idx = np.empty((K1, 2), np.int32)
    d2 = np.empty((K1, 2), np.float64)
    lib().orc_knn2(q, K1, t.reshape(-1), K2, D, idx, d2)
    return idx, d2


def match_pair(q, t, ratio=RATIO):
    """out[i] = train index matched to query i, or -1 (FlannMatcher::matchFeatures)."""
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    K1 = q.shape[0]
    D = q.shape[1] if q.ndim == 2 else t.shape[1]
    out = np.full(max(K1, 1), -1, np.int32)
    n = lib().orc_match_pair(q.reshape(-1), K1, t.reshape(-1), t.shape[0], D,
                             float(ratio), out)
    return out[:K1], n


def match_grid(images, pairs, ratio=RATIO, threads=0):
    """images: list of (K_n, D) float32 arrays; pairs: (P,2) int (query image, train image).

    Returns (out[P, Kmax] int32 with -1 padding, counts[P]).
    """
    D = images[0].shape[1]
    rows = np.array([0] + [im.shape[0] for im in images], np.int64).cumsum()
    desc = np.ascontiguousarray(np.concatenate([im.reshape(-1, D) for im in images], 0),
                                np.float32)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    P = pairs.shape[0]
    kmax = max(1, max(im.shape[0] for im in images))
    out = np.full((P, kmax), -1, np.int32)
    counts = np.zeros(P, np.int32)
    lib().orc_match_grid(desc.reshape(-1), rows, len(images), D, pairs.reshape(-1), P,
                         float(ratio), out.reshape(-1), kmax, counts, threads)
    return out, counts


def desc_sample(desc_map, kp_xy, D=256, channel_last=False):
    """FeatureSuperPoint::processDescriptors: desc_map [C][Hc][Wc] (the network's layout) or [Hc][Wc][C];
    kp_xy (K, 2) integer image coordinates (x, y).  Returns (K, D) fp32 unit-norm rows."""
    m = np.ascontiguousarray(desc_map, np.float32)
    kp = np.ascontiguousarray(kp_xy, np.int32).reshape(-1, 2)
    if channel_last:
        Hc, Wc, Cn = m.shape
        sc, sy, sx = 1, Wc * Cn, Cn
    else:
        Cn, Hc, Wc = m.shape
        sc, sy, sx = Hc * Wc, Wc, 1
    out = np.zeros((len(kp), D), np.float32)
    lib().orc_desc_sample(m.reshape(-1), sc, sy, sx, kp.reshape(-1), len(kp), D, out.reshape(-1))
    return out


def all_pairs(n):
    """Canonical pair list: for i<j, query=i, train=j (SequentialReconstructor.cpp:202-227:
    the (j,i) entry is the inverted map of (i,j) and is not matched again)."""
    return np.array([(i, j) for i in range(n) for j in range(i + 1, n)], np.int32).reshape(-1, 2)

"""Independent re-statements used to cross-check the C oracle (tests only).

exact_*: pure Python, every fp64 fma emulated with exact rationals (small cases only).
numpy_*: float64 numpy (pairwise summation: identical indices except on last-ulp near ties,
         exact on integer-valued descriptors).
"""
from fractions import Fraction

import numpy as np

F32 = np.float32


def exact_l2sq(q, t):
    acc = 0.0
    for a, b in zip(q, t):
        d = float(a) - float(b)              # IEEE double subtraction
        acc = float(Fraction(d) * Fraction(d) + Fraction(acc))  # one rounding = fma
    return acc


def ratio_unique(idx0, d0, d1, K2, ratio):
    ratio = F32(ratio)
    out = np.full(len(idx0), -1, np.int32)
    taken = set()
    for i in range(len(idx0)):
        if K2 < 2:
            continue
        dist0 = np.sqrt(F32(d0[i]))
        dist1 = np.sqrt(F32(d1[i]))
        if dist0 < ratio * dist1 and idx0[i] not in taken:   # FeatureMatcher.cpp:55,58
            taken.add(int(idx0[i]))
            out[i] = idx0[i]
    return out


def exact_match_pair(q, t, ratio=0.7):
    K1, K2 = len(q), len(t)
    i0 = np.zeros(K1, np.int64); d0 = np.full(K1, np.inf); d1 = np.full(K1, np.inf)
    for i in range(K1):
        ds = sorted((exact_l2sq(q[i], t[j]), j) for j in range(K2))
        if K2 >= 1:
            d0[i], i0[i] = ds[0]
        if K2 >= 2:
            d1[i] = ds[1][0]
    return ratio_unique(i0, d0, d1, K2, ratio)


def numpy_match_pair(q, t, ratio=0.7):
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    K1, K2 = len(q), len(t)
    if K1 == 0:
        return np.zeros(0, np.int32)
    if K2 < 2:
        return np.full(K1, -1, np.int32)
    d2 = ((q64[:, None, :] - t64[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind="stable")
    r = np.arange(K1)
    return ratio_unique(order[:, 0], d2[r, order[:, 0]], d2[r, order[:, 1]], K2, ratio)

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


# ---- landmark validity sweep: literal transcription of the control flow with Python lists -----
def validity_python(poses34, intr, points, pt_off, obs_cam, obs_xy, max_err=4.0, min_angle=1.0):
    """checkLandmarkValidity with an actual list and `del`, numpy float64 scalars (round-to-nearest,
    one operation at a time).  Slow: small cases only."""
    import math
    poses34 = np.asarray(poses34, np.float64).reshape(-1, 3, 4)
    inl, keep = [], np.zeros(len(obs_cam), bool)
    for j in range(len(points)):
        X = np.asarray(points[j], np.float64)
        track = list(range(pt_off[j], pt_off[j + 1]))
        ok = True
        i = 0
        while i < len(track):
            o = track[i]
            P, K = poses34[obs_cam[o]], intr[obs_cam[o]]
            l = [((P[r, 0] * X[0] + P[r, 1] * X[1]) + P[r, 2] * X[2]) + P[r, 3] for r in range(3)]
            with np.errstate(all="ignore"):
                x, y = np.float64(l[0]) / np.float64(l[2]), np.float64(l[1]) / np.float64(l[2])
                rad = x * x + y * y
                dist = K[4] * rad + (K[5] * rad) * rad
                u, v = K[0] * (x + dist) + K[2], K[1] * (y + dist) + K[3]
                resid = abs(u - float(obs_xy[o][0])) + abs(v - float(obs_xy[o][1]))
            if resid > max_err or l[2] < 0:
                del track[i]
                if len(track) < 2:
                    ok = False
            i += 1
        angle_ok = False
        for a in track:
            for b in track:
                if a == b:
                    continue
                Pa, Pb = poses34[obs_cam[a]], poses34[obs_cam[b]]
                ca = [((-Pa[0, c]) * Pa[0, 3] + (-Pa[1, c]) * Pa[1, 3]) + (-Pa[2, c]) * Pa[2, 3] for c in range(3)]
                cb = [((-Pb[0, c]) * Pb[0, 3] + (-Pb[1, c]) * Pb[1, 3]) + (-Pb[2, c]) * Pb[2, 3] for c in range(3)]
                r1 = [X[c] - ca[c] for c in range(3)]
                r2 = [X[c] - cb[c] for c in range(3)]
                dot = (r1[0] * r2[0] + r1[1] * r2[1]) + r1[2] * r2[2]
                n1 = np.sqrt((r1[0] * r1[0] + r1[1] * r1[1]) + r1[2] * r1[2])
                n2 = np.sqrt((r2[0] * r2[0] + r2[1] * r2[1]) + r2[2] * r2[2])
                with np.errstate(all="ignore"):
                    cosv = np.float64(dot) / (n1 * n2)
                ang = 180.0 * math.acos(cosv) / 3.1415 if -1.0 <= cosv <= 1.0 else float("nan")
                if ang > min_angle:
                    angle_ok = True
        if not angle_ok:
            ok = False
        inl.append(ok)
        keep[track] = True
    return np.array(inl, bool), keep

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


# ---- epipolar filter: same control flow, different numerics (SVD null space, numpy.roots) --------
def fmat_python(xy1, xy2):
    """findFundamentalMat's default path re-derived with numpy linear algebra: the sampling sequence
    and the accept logic are transcribed, the 7-point models come from an SVD null space and
    numpy.roots (ordered smallest, largest, middle like the closed form).  Returns (mask, count, iters)."""
    import math
    m1 = np.asarray(xy1, np.float64).reshape(-1, 2)
    m2 = np.asarray(xy2, np.float64).reshape(-1, 2)
    n = len(m1)
    if n < 7:
        return np.ones(n, bool), -2, 0
    state = [(1 << 64) - 1]

    def nxt():
        s = state[0]
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & ((1 << 64) - 1)
        state[0] = s
        return s & 0xFFFFFFFF

    def collinear(m):
        i = 6
        for j in range(i):
            dx1, dy1 = m[j] - m[i]
            for k in range(j):
                dx2, dy2 = m[k] - m[i]
                if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                    return True
        return False

    def draw():
        for _ in range(10000):
            idx = []
            while len(idx) < 7:
                v = nxt() % n
                if v not in idx:
                    idx.append(v)
            if not collinear(m1[idx]) and not collinear(m2[idx]):
                return idx
        return None

    def seven(idx):
        A = np.array([[x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0]
                      for (x0, y0), (x1, y1) in zip(m1[idx], m2[idx])])
        _, sv, vt = np.linalg.svd(A)
        if sv[6] <= 1e-12 * sv[0]:
            return []
        f1, f2 = vt[7].reshape(3, 3), vt[8].reshape(3, 3)
        lam = np.array([-1.0, 0.0, 1.0, 2.0])
        coef = np.polyfit(lam, [np.linalg.det(l * f1 + (1 - l) * f2) for l in lam], 3)
        roots = np.roots(coef)
        real = sorted(r.real for r in roots if abs(r.imag) < 1e-9 * max(1.0, abs(r.real)))
        if len(real) == 3:
            real = [real[0], real[2], real[1]]
        out = []
        for l in real:
            F = l * f1 + (1 - l) * f2
            out.append(F / F[2, 2] if abs(F[2, 2]) > 1e-300 else F)
        return out

    def errors(F):
        h1 = np.c_[m1, np.ones(n)]
        h2 = np.c_[m2, np.ones(n)]
        l2 = h1 @ F.T
        l1 = h2 @ F
        d2 = (h2 * l2).sum(1) ** 2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
        d1 = (h1 * l1).sum(1) ** 2 / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
        return np.maximum(d1, d2).astype(np.float32)

    def niters_for(ep, max_iters):
        num = max(1 - 0.99, 2.2250738585072014e-308)
        den = 1 - (1 - ep) ** 7
        if den < 2.2250738585072014e-308:
            return 0
        num, den = math.log(num), math.log(den)
        return max_iters if den >= 0 or -num >= max_iters * (-den) else int(round(num / den))   # banker's rounding = cvRound

    if n == 7:
        ok = len(seven(list(range(7)))) > 0
        return np.full(7, ok), (7 if ok else -1), 1
    best, done = None, 0
    if n >= 15:
        niters, max_good, mask = 1000, 0, np.zeros(n, bool)
        it = 0
        while it < niters:
            idx = draw()
            if idx is None:
                break
            done = it + 1
            for F in seven(idx):
                cur = errors(F) <= 9.0
                g = int(cur.sum())
                if g > max(max_good, 6):
                    mask, max_good = cur, g
                    niters = niters_for((n - g) / n, niters)
            it += 1
        return (mask, max_good, done) if max_good > 0 else (np.zeros(n, bool), -1, done)
    niters = max(niters_for(0.45, 1000), 3)
    min_med = float("inf")
    for it in range(niters):
        idx = draw()
        if idx is None:
            break
        done = it + 1
        for F in seven(idx):
            e = np.sort(errors(F))
            med = float(e[n // 2]) if n % 2 else float(np.float32(e[n // 2 - 1] + e[n // 2])) * 0.5
            if med < min_med:
                min_med, best = med, F
    if best is None:
        return np.zeros(n, bool), -1, done
    sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * math.sqrt(min_med), 0.001)
    mask = errors(best) <= sigma * sigma
    g = int(mask.sum())
    return (mask, g, done) if g >= 7 else (np.zeros(n, bool), -1, done)

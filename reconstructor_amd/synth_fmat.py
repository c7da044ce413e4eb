"""Synthetic two-view correspondences for the epipolar filter: seeded pinhole pairs observing a
random point cloud, pixel noise, truncation to integer coordinates (the reference's features carry
ints, FeatureDetector.cpp:28-29) and a share of gross outliers."""
import numpy as np


def two_view(n, outlier_frac=0.3, seed=0, noise_px=0.5, width=512, height=336):
    """(xy1, xy2, is_outlier): n integer correspondences between two views of one scene."""
    r = np.random.default_rng(seed)
    f = 1.2 * max(width, height)
    K = np.array([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1.0]])
    X = np.c_[r.uniform(-2, 2, (n, 2)), r.uniform(4, 9, n)]
    th, ph = r.uniform(-0.3, 0.3), r.uniform(-0.1, 0.1)
    Ry = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(ph), -np.sin(ph)], [0, np.sin(ph), np.cos(ph)]])
    t = r.uniform(-1, 1, 3) * np.array([1.0, 0.3, 0.3])
    p1 = (K @ X.T).T
    p1 = p1[:, :2] / p1[:, 2:]
    X2 = (Rx @ Ry @ X.T).T + t
    p2 = (K @ X2.T).T
    p2 = p2[:, :2] / p2[:, 2:]
    p1 = p1 + r.normal(0, noise_px, p1.shape)
    p2 = p2 + r.normal(0, noise_px, p2.shape)
    bad = r.random(n) < outlier_frac
    p2[bad] = np.c_[r.uniform(0, width, bad.sum()), r.uniform(0, height, bad.sum())]
    return np.trunc(p1).astype(np.int32), np.trunc(p2).astype(np.int32), bad


def grid(sizes, outlier_frac=0.3, seed=0):
    """CSR batch of independent pairs with the given match counts: (pair_off, xy1, xy2)."""
    off, a, b = [0], [], []
    for p, n in enumerate(sizes):
        x, y, _ = two_view(int(n), outlier_frac, seed=seed * 100003 + p) if n > 0 else (np.zeros((0, 2), np.int32),) * 2 + (None,)
        a.append(x); b.append(y); off.append(off[-1] + int(n))
    cat = lambda v: np.concatenate(v).reshape(-1, 2) if len(v) else np.zeros((0, 2), np.int32)
    return np.asarray(off, np.int32), cat(a), cat(b)

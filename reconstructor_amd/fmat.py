"""Host mirror of the reference's epipolar match filter over librcn.so (no CPU fallback).

    GeometricFilter::estimateFundamental                     GeometricFilter.cpp:39-61
    its use inside SequentialReconstructor::matchFeatures    SequentialReconstructor.cpp:237-269
"""
import ctypes as C

import numpy as np


def estimate_fundamental_inliers(ctx, xy1, xy2, with_matrix=False):
    """One pair: (mask[n] bool, count[, F 3x3]).  count -1: no model, -2: fewer than 7 points (unfiltered)."""
    xy1 = np.ascontiguousarray(xy1, np.int32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.int32).reshape(-1, 2)
    if len(xy1) != len(xy2):
        raise ValueError("one point of the second image per point of the first")
    mask = np.zeros(len(xy1), np.uint8)
    cnt = C.c_int32(0)
    F = np.zeros((3, 3))
    ctx.check(ctx.lib.rcn_fmat_filter(ctx.h, xy1.ctypes.data, xy2.ctypes.data, len(xy1), mask.ctypes.data, C.addressof(cnt),
                                      F.ctypes.data if with_matrix else None))
    return (mask.astype(bool), cnt.value, F) if with_matrix else (mask.astype(bool), cnt.value)


def filter_grid(ctx, pair_off, xy1, xy2):
    """CSR batch: (mask[N] bool, counts[P], iterations[P])."""
    pair_off = np.ascontiguousarray(pair_off, np.int32)
    xy1 = np.ascontiguousarray(xy1, np.int32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.int32).reshape(-1, 2)
    P = len(pair_off) - 1
    mask = np.zeros(len(xy1), np.uint8)
    counts = np.zeros(P, np.int32)
    iters = np.zeros(P, np.int32)
    ctx.check(ctx.lib.rcn_fmat_filter_grid(ctx.h, P, pair_off.ctypes.data, xy1.ctypes.data, xy2.ctypes.data,
                                           mask.ctypes.data, counts.ctypes.data, iters.ctypes.data, None))
    return mask.astype(bool), counts, iters


def matches_to_csr(coords, pairs, table):
    """The point lists the pair loop hands to estimateFundamental (:240-248): for every pair, the
    matched features in ascending query order.  coords: list of (K_i, 2) int arrays (featCoord);
    table: the matcher's output, table[p, q] = train feature or -1."""
    off, a, b = [0], [], []
    for p, (i, j) in enumerate(np.asarray(pairs)):
        q = np.flatnonzero(table[p, :len(coords[i])] >= 0)
        a.append(np.asarray(coords[i], np.int32)[q])
        b.append(np.asarray(coords[j], np.int32)[table[p, q]])
        off.append(off[-1] + len(q))
    cat = lambda v: np.concatenate(v) if v else np.zeros((0, 2), np.int32)
    return np.asarray(off, np.int32), cat(a).reshape(-1, 2), cat(b).reshape(-1, 2)


def apply_geometric_filter(ctx, coords, pairs, table):
    """SequentialReconstructor.cpp:237-269 on a whole match table: pairs with >= 7 matches keep only
    the inliers (none when no model is found), the others are left as they are."""
    off, xy1, xy2 = matches_to_csr(coords, pairs, table)
    mask, counts, _ = filter_grid(ctx, off, xy1, xy2)
    out = table.copy()
    for p, (i, j) in enumerate(np.asarray(pairs)):
        q = np.flatnonzero(table[p, :len(coords[i])] >= 0)
        if len(q) >= 7:
            out[p, q[~mask[off[p]:off[p + 1]]]] = -1
    return out, counts

"""Fingerprint of a match table, used to pin whole grids against the CPU oracle without storing them:
per pair, a 64-bit FNV-1a-style hash over the row's int32 entries (h = (h ^ entry) * 1099511628211 mod 2^64,
entry taken as an unsigned 32-bit word, starting from 14695981039346656037) and the match count.
Pure numpy, vectorised over the pairs; the same function hashes the oracle's table (tests/golden/make_*_golden.py)
and the table the GPU left (tests, bench.py)."""
import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def row_hashes(table, width=None):
    """table: (P, >= width) int32, out[p, q] = train row or -1.  Returns (hash[P] uint64, count[P] int32) over the
    first `width` entries of every row."""
    t = np.ascontiguousarray(table, np.int32)
    if width is not None:
        t = t[:, :width]
    u = t.view(np.uint32) if t.flags.c_contiguous else np.ascontiguousarray(t).view(np.uint32)
    h = np.full(t.shape[0], FNV_OFFSET, np.uint64)
    with np.errstate(over="ignore"):
        for k in range(t.shape[1]):
            h = (h ^ u[:, k].astype(np.uint64)) * FNV_PRIME
    return h, (t >= 0).sum(1).astype(np.int32)

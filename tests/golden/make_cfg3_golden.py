"""Writes tests/golden/match_cfg3_sample.npz: oracle outputs for 64 image pairs sampled from BASELINE.json
configs[2] (1000 images x 4096 SuperPoint-like 256-d keypoints, seed 1234 -- the set bench.py matches), and
tests/golden/match_cfg3_sample512.npz: row hash + match count (reconstructor_amd/tablehash.py) of 512 more pairs,
64 for every residue of the pair number modulo 8 -- the deal of the canonical grid to the ranks of an 8-GPU node
(pair number p -> rank p % 8), so every rank's share is sampled.

  python tests/golden/make_cfg3_golden.py      (about a minute of CPU work; run in the build container)
  python tests/golden/make_cfg3_golden.py --5k (round 5: match_cfg3_sample5k.npz, 5120 more pairs by row hash + count, 40 for every
                                                sixteenth of the pair list x residue modulo 8; about ten minutes on 7 threads)

Stored per pair of the first file: the (query row, train row) lists in ascending query order and the match count.
The GPU test (tests/test_cfg3_gpu.py) runs the WHOLE 499 500-pair grid and compares these rows outright.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc                      # noqa: E402
from reconstructor_amd import synth, tablehash        # noqa: E402

N, K, SEED = 1000, 4096, 1234


def sample_pairs():
    rng = np.random.default_rng(20261004)
    fixed = [(0, 1), (0, 999), (998, 999), (499, 500), (7, 8), (123, 877)]
    seen = set(fixed)
    while len(seen) < 64:
        i, j = sorted(int(x) for x in rng.integers(0, N, 2))
        if i != j:
            seen.add((i, j))
    return np.array(sorted(seen), np.int32)


def pair_of(p, n=N):
    """(i, j) of pair number p in the canonical row-major i < j list."""
    i = int((2 * n - 1 - np.sqrt((2 * n - 1) ** 2 - 8 * p)) // 2)
    while i * (2 * n - i - 1) // 2 > p:
        i -= 1
    while (i + 1) * (2 * n - i - 2) // 2 <= p:
        i += 1
    return i, int(p - i * (2 * n - i - 1) // 2 + i + 1)


def sample_pairs_512():
    rng = np.random.default_rng(20261005)
    total = N * (N - 1) // 2
    numbers = []
    for r in range(8):
        cand = rng.choice((total - r + 7) // 8, size=64, replace=False) * 8 + r
        numbers += [int(x) for x in cand if x < total]
    numbers = sorted(set(numbers))
    return np.array(numbers, np.int64), np.array([pair_of(p) for p in numbers], np.int32)


def hashed_sample():
    numbers, pairs = sample_pairs_512()
    assert len(pairs) == 512 and all(np.bincount(numbers % 8, minlength=8) == 64)
    ids = sorted(set(pairs.reshape(-1).tolist()))
    pool = synth.world_pool("superpoint", 4 * K, seed=SEED)
    ims = [synth.image_descriptors("superpoint", i, K, pool, seed=SEED) for i in ids]
    remap = {g: l for l, g in enumerate(ids)}
    local = np.array([(remap[a], remap[b]) for a, b in pairs], np.int32)
    out, counts = orc.match_grid(ims, local, threads=8)
    h, c = tablehash.row_hashes(out, K)
    assert np.array_equal(c, counts)
    path = os.path.join(ROOT, "tests", "golden", "match_cfg3_sample512.npz")
    np.savez_compressed(path, pair_numbers=numbers, pairs=pairs, hashes=h, counts=counts, n_images=N, K=K, seed=SEED)
    print(path, os.path.getsize(path), "bytes;", len(pairs), "pairs,", int(counts.sum()), "matches")


def sample_pairs_5k(per_stratum=40, n_chunks=16):
    """Round 5: 5120 pairs, 40 for every (sixteenth of the canonical pair list) x (residue of the pair number modulo 8):
    the sixteenths stand for the pipeline chunks of the one-GPU grid call (chunks are consecutive ranges of the list; their exact
    borders depend on the workspace budget, every sixteenth overlaps at most two of them), the residues for the ranks of an 8-GPU node."""
    rng = np.random.default_rng(20261006)
    total = N * (N - 1) // 2
    numbers = []
    for c in range(n_chunks):
        lo, hi = total * c // n_chunks, total * (c + 1) // n_chunks
        for r in range(8):
            first = lo + ((r - lo) % 8)
            cnt = (hi - first + 7) // 8
            pick = rng.choice(cnt, size=per_stratum, replace=False)
            numbers += [int(first + 8 * x) for x in pick]
    numbers = sorted(set(numbers))
    return np.array(numbers, np.int64), np.array([pair_of(p) for p in numbers], np.int32)


def hashed_sample_5k(threads=7):
    numbers, pairs = sample_pairs_5k()
    assert len(pairs) == 5120
    total = N * (N - 1) // 2
    strata = np.bincount((numbers * 16 // total) * 8 + numbers % 8, minlength=128)
    assert (strata == 40).all()
    ids = sorted(set(pairs.reshape(-1).tolist()))
    pool = synth.world_pool("superpoint", 4 * K, seed=SEED)
    ims = [synth.image_descriptors("superpoint", i, K, pool, seed=SEED) for i in ids]
    remap = {g: l for l, g in enumerate(ids)}
    local = np.array([(remap[a], remap[b]) for a, b in pairs], np.int32)
    hs, cs = [], []
    for s0 in range(0, len(local), 256):          # in slices: the oracle's output table is K ints per pair
        out, counts = orc.match_grid(ims, local[s0:s0 + 256], threads=threads)
        h, c = tablehash.row_hashes(out, K)
        assert np.array_equal(c, counts)
        hs.append(h); cs.append(counts)
        print("  %d / %d pairs" % (min(s0 + 256, len(local)), len(local)), flush=True)
    h, counts = np.concatenate(hs), np.concatenate(cs)
    path = os.path.join(ROOT, "tests", "golden", "match_cfg3_sample5k.npz")
    np.savez_compressed(path, pair_numbers=numbers, pairs=pairs, hashes=h, counts=counts, n_images=N, K=K, seed=SEED)
    print(path, os.path.getsize(path), "bytes;", len(pairs), "pairs,", int(counts.sum()), "matches")


def main():
    if "--5k" in sys.argv:
        hashed_sample_5k()
        return
    if "--only-64" not in sys.argv:
        hashed_sample()
    if "--only-512" in sys.argv:
        return
    pairs = sample_pairs()
    ids = sorted(set(pairs.reshape(-1).tolist()))
    pool = synth.world_pool("superpoint", 4 * K, seed=SEED)
    ims = [synth.image_descriptors("superpoint", i, K, pool, seed=SEED) for i in ids]
    remap = {g: l for l, g in enumerate(ids)}
    local = np.array([(remap[a], remap[b]) for a, b in pairs], np.int32)
    out, counts = orc.match_grid(ims, local, threads=8)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    qt = np.zeros((offs[-1], 2), np.int32)
    for p in range(len(pairs)):
        q = np.nonzero(out[p] >= 0)[0]
        qt[offs[p]:offs[p + 1], 0] = q
        qt[offs[p]:offs[p + 1], 1] = out[p][q]
    path = os.path.join(ROOT, "tests", "golden", "match_cfg3_sample.npz")
    np.savez_compressed(path, pairs=pairs, counts=counts, offsets=offs, qt=qt.astype(np.int16),
                        n_images=N, K=K, seed=SEED)
    print(path, os.path.getsize(path), "bytes;", len(pairs), "pairs,", int(offs[-1]), "matches")


if __name__ == "__main__":
    main()

"""Writes tests/golden/match_cfg3_sample.npz: oracle outputs for 64 image pairs sampled from BASELINE.json
configs[2] (1000 images x 4096 SuperPoint-like 256-d keypoints, seed 1234 -- the set bench.py matches).

  python tests/golden/make_cfg3_golden.py      (about 10 s of CPU work; run in the build container)

Stored per sampled pair: the (query row, train row) lists in ascending query order and the match count.
The GPU test (tests/test_cfg3_gpu.py) runs the WHOLE 499 500-pair grid and compares these 64 rows outright.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc                      # noqa: E402
from reconstructor_amd import synth        # noqa: E402

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


def main():
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

"""Writes tests/golden/match_cfg2_full.npz: the CPU oracle's result for ALL 4950 image pairs of BASELINE.json
configs[1] (100 images x 2048 SuperPoint-like 256-d keypoints, seed 1234 -- the set bench.py's cfg2 leg matches),
as one 64-bit row hash + match count per pair (reconstructor_amd/tablehash.py).

  python tests/golden/make_cfg2_golden.py      (2.1e10 exact fp64 pair-distances: a few minutes on 8 cores)

The GPU test (tests/test_match_gpu.py::test_cfg2_full_grid_equals_the_oracle) hashes the table the GPU leaves and
compares all 4950 pairs; bench.py does the same on the table of its cfg2 leg (`cfg2.equal_to_cpu`).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc                                   # noqa: E402
from reconstructor_amd import synth, tablehash          # noqa: E402

N, K, SEED = 100, 2048, 1234


def main():
    pool = synth.world_pool("superpoint", 4 * K, seed=SEED)
    ims = [synth.image_descriptors("superpoint", i, K, pool, seed=SEED) for i in range(N)]
    i, j = np.triu_indices(N, 1)
    pairs = np.stack([i, j], 1).astype(np.int32)
    t0 = time.perf_counter()
    out, counts = orc.match_grid(ims, pairs, threads=0)
    dt = time.perf_counter() - t0
    h, c = tablehash.row_hashes(out, K)
    assert np.array_equal(c, counts)
    path = os.path.join(ROOT, "tests", "golden", "match_cfg2_full.npz")
    np.savez_compressed(path, pairs=pairs, hashes=h, counts=counts, n_images=N, K=K, seed=SEED,
                        matches_found=np.int64(counts.sum()), oracle_seconds=dt)
    print(path, os.path.getsize(path), "bytes;", len(pairs), "pairs,", int(counts.sum()), "matches, oracle %.1f s" % dt)


if __name__ == "__main__":
    main()

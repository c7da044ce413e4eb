#!/usr/bin/env python3
"""tools/soak_match.py [seconds] [seed] -- randomised differential run of the matcher against the CPU oracle:
random image counts, ragged keypoint counts (1 .. 3000, a few beyond 8192), the three descriptor kinds, duplicated and
near-duplicated rows, random pair lists with repeated and reversed pairs.  Every table must equal the oracle's bit for bit.
Not part of the test-suite (minutes of GPU + CPU time); run on the MI355X box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from oracle import orc
    from reconstructor_amd import synth
    from reconstructor_amd.matcher import HipL2Matcher
    rng = np.random.default_rng(seed)
    m = HipL2Matcher(device=0)
    t0, cases, rows = time.time(), 0, 0
    t_said = time.time()
    while time.time() - t0 < budget:
        if time.time() - t_said > 60.0:      # (a run that says nothing for minutes is taken for hung)
            t_said = time.time()
            print("  ... %.0f s" % (time.time() - t0), flush=True)
        kind = rng.choice(["superpoint", "superpoint", "superpoint", "sift", "orb"])
        n = int(rng.integers(2, 7))
        big = rng.random() < 0.05
        Ks = [int(rng.integers(8200, 12000)) if (big and i < 2) else int(rng.choice([1, 2, 3, 17, 63, 64, 65, 511, 512, 513, int(rng.integers(1, 3000))]))
              for i in range(n)]
        ims = synth.descriptor_set(kind, n, Ks, n_world=int(rng.integers(64, 6000)), seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.5:      # exact duplicates and near-duplicates inside an image: ties and near-ties for the top-2
            for im in ims:
                if len(im) >= 4:
                    a, b = rng.integers(0, len(im), 2)
                    im[a] = im[b]
                    c, d = rng.integers(0, len(im), 2)
                    im[c] = im[d] + (rng.standard_normal(im.shape[1]) * 1e-4).astype(np.float32)
        P = int(rng.integers(1, 2 * n * n))
        pairs = rng.integers(0, n, (P, 2)).astype(np.int32)
        pairs = pairs[pairs[:, 0] != pairs[:, 1]]
        if len(pairs) == 0:
            continue
        ratio = float(rng.choice([0.7, 0.7, 0.8, 0.5, 0.95]))
        m.clear()
        for i, im in enumerate(ims):
            m.upload(i, im)
        kmax = max(Ks)
        m.ratio = float(np.float32(ratio))
        out, cnt = m.match_grid(pairs, kmax)
        exp, ec = orc.match_grid(ims, pairs, ratio=ratio, threads=16)
        if not (np.array_equal(out, exp) and np.array_equal(cnt, ec)):
            bad = np.argwhere(out != exp)
            print("MISMATCH", kind, Ks, pairs.tolist(), ratio, bad[:5].tolist(), flush=True)
            sys.exit(1)
        cases += 1
        rows += int(sum(Ks[a] for a in pairs[:, 0]))
    print("soak ok: %d grids, %d query rows, %.0f s, seed %d" % (cases, rows, time.time() - t0, seed))


if __name__ == "__main__":
    main()

"""Generates tests/golden/match_*.npz: inputs + expected outputs of the matcher path.

The reference holds no fixtures for this path, so the vectors are made
here: expected maps come from the C oracle (oracle/match_oracle.c) and every one is
cross-checked against an independent computation (tests/indep.py: exact-rational Python for
the known-answer cases, float64 numpy for the seeded sets) before it is written.
Run from the repo root:  python tests/golden/make_match_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import indep  # noqa: E402
from oracle import orc  # noqa: E402
from reconstructor_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def kat_cases():
    cases = {}
    rng = np.random.default_rng(7)
    # 1. exact tie for best between train rows 1 and 3 -> ratio fails (d0 == d1)
    q = rng.standard_normal((4, 8)).astype(F)
    t = rng.standard_normal((5, 8)).astype(F)
    t[3] = t[1]
    t[1] = q[0] + F(0.25)
    t[3] = t[1]
    cases["tie_best"] = (q, t)
    # 2. exact tie for SECOND best only: best is unique and close -> lower index second, passes
    q = np.zeros((1, 4), F)
    t = np.array([[3, 0, 0, 0], [0.5, 0, 0, 0], [0, 3, 0, 0], [0, 0, 9, 0]], F)
    cases["tie_second"] = (q, t)
    # 3. duplicate queries claim the same train row: the lower query index keeps it
    t = rng.standard_normal((6, 16)).astype(F) * 4
    q = np.stack([t[2] + F(0.01), t[2] + F(0.01), t[4] - F(0.02), t[2] + F(0.011)]).astype(F)
    cases["dup_query"] = (q, t)
    # 4. K2 == 2, K2 == 1, K2 == 0, K1 == 0
    q = rng.standard_normal((3, 8)).astype(F)
    t2 = np.stack([q[1] + F(0.001), q[1] + 5]).astype(F)
    cases["k2_is_2"] = (q, t2)
    cases["k2_is_1"] = (q, t2[:1])
    cases["k2_is_0"] = (q, np.zeros((0, 8), F))
    cases["k1_is_0"] = (np.zeros((0, 8), F), t2)
    # 5. ratio exactly at the threshold: dist0 == 0.7f * dist1 -> strict < fails; one ulp
    #    below passes
    a = F(0.7)
    below = np.nextafter(a, F(0))
    q = np.zeros((2, 4), F)
    q[1, 1] = 100  # second query far away in another axis
    t = np.array([[a, 0, 0, 0], [1, 0, 0, 0], [0, 100 + below, 0, 0], [0, 101, 0, 0]], F)
    cases["ratio_equal"] = (q, t)
    # 6. collision chain: queries 0..2 all want train 5; 1 and 2 get nothing
    t = (rng.standard_normal((8, 12)) * 3).astype(F)
    q = np.stack([t[5] + F(0.03), t[5] - F(0.02), t[5] + F(0.01), t[0] + F(0.01)]).astype(F)
    cases["collision_chain"] = (q, t)
    # 7. integer-valued (ORB-as-float) with many exact ties
    q = rng.integers(0, 3, (12, 32)).astype(F)
    t = rng.integers(0, 3, (20, 32)).astype(F)
    t[7] = q[3]
    t[9] = q[3]
    cases["int_ties"] = (q, t)
    # 8. zero descriptors and a huge dynamic range
    q = np.zeros((3, 8), F)
    q[1] = 1e-20
    q[2] = 1e18
    t = np.zeros((4, 8), F)
    t[1] = 1e18
    t[2, 0] = 1e-20
    t[3] = -1e18
    cases["zeros_range"] = (q, t)
    return cases


def main():
    kats = {}
    for name, (q, t) in kat_cases().items():
        exp, n = orc.match_pair(q, t) if q.shape[0] else (np.zeros(0, np.int32), 0)
        ind = indep.exact_match_pair(q, t)
        assert np.array_equal(exp, ind), (name, exp, ind)
        kats[name + "/q"], kats[name + "/t"], kats[name + "/expect"] = q, t, exp
        print("KAT %-16s K1=%d K2=%d D=%d matches=%d" % (name, q.shape[0], t.shape[0],
                                                        q.shape[1], (exp >= 0).sum()))
    np.savez_compressed(os.path.join(OUT, "match_kat.npz"), **kats)

    seeded = {}
    for kind, K1, K2 in (("superpoint", 96, 80), ("sift", 70, 120), ("orb", 90, 64)):
        ims = synth.descriptor_set(kind, 2, [K1, K2], n_world=160, seed=99)
        q, t = ims
        exp, n = orc.match_pair(q, t)
        ind = indep.numpy_match_pair(q, t)
        assert np.array_equal(exp, ind), kind
        if kind == "orb":   # also exact-rational on a slice
            assert np.array_equal(indep.exact_match_pair(q[:10], t), orc.match_pair(q[:10], t)[0])
        seeded[kind + "/q"], seeded[kind + "/t"], seeded[kind + "/expect"] = q, t, exp
        print("seeded %-10s K1=%d K2=%d matches=%d" % (kind, K1, K2, n))
    np.savez_compressed(os.path.join(OUT, "match_seeded.npz"), **seeded)

    # 8-image ragged grid, D=256 (inputs are regenerated from the seed; only outputs stored)
    ks = [64, 96, 80, 128, 72, 100, 64, 90]
    ims = synth.descriptor_set("superpoint", 8, ks, n_world=256, seed=4242)
    pairs = orc.all_pairs(8)
    out, counts = orc.match_grid(ims, pairs, threads=4)
    for p, (a, b) in enumerate(pairs):
        ind = indep.numpy_match_pair(ims[a], ims[b])
        assert np.array_equal(out[p, :ks[a]], ind), (a, b)
    np.savez_compressed(os.path.join(OUT, "match_grid8.npz"), ks=np.array(ks), seed=4242,
                        n_world=256, pairs=pairs, expect=out, counts=counts)
    print("grid8: %d pairs, %d matches" % (len(pairs), counts.sum()))


if __name__ == "__main__":
    main()

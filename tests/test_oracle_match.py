"""CPU suite: the matcher oracle against the committed golden vectors and against
independent re-statements (no GPU)."""
import os

import numpy as np
import pytest

import indep
from oracle import orc
from reconstructor_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def _cases(npz):
    z = np.load(os.path.join(G, npz))
    names = sorted({k.split("/")[0] for k in z.files})
    return [(n, z[n + "/q"], z[n + "/t"], z[n + "/expect"]) for n in names]


@pytest.mark.parametrize("name,q,t,expect", _cases("match_kat.npz"), ids=lambda v: v if isinstance(v, str) else "")
def test_kat(name, q, t, expect):
    got, n = orc.match_pair(q, t) if q.shape[0] else (np.zeros(0, np.int32), 0)
    assert np.array_equal(got, expect)
    assert n == (expect >= 0).sum()
    assert np.array_equal(indep.exact_match_pair(q, t), expect)


@pytest.mark.parametrize("name,q,t,expect", _cases("match_seeded.npz"), ids=lambda v: v if isinstance(v, str) else "")
def test_seeded(name, q, t, expect):
    got, _ = orc.match_pair(q, t)
    assert np.array_equal(got, expect)
    assert np.array_equal(indep.numpy_match_pair(q, t), expect)


def test_grid8_matches_golden_and_inverse_pairs():
    z = np.load(os.path.join(G, "match_grid8.npz"))
    ims = synth.descriptor_set("superpoint", 8, list(z["ks"]), n_world=int(z["n_world"]), seed=int(z["seed"]))
    out, counts = orc.match_grid(ims, z["pairs"], threads=2)
    assert np.array_equal(out, z["expect"])
    assert np.array_equal(counts, z["counts"])
    # per-pair entry point agrees with the grid entry point
    a, b = z["pairs"][5]
    assert np.array_equal(orc.match_pair(ims[a], ims[b])[0], out[5, :ims[a].shape[0]])


def test_l2sq_is_the_fma_chain():
    rng = np.random.default_rng(3)
    for D in (1, 7, 32, 128, 256):
        q = (rng.standard_normal(D) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        t = (rng.standard_normal(D) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        assert orc.l2sq(q, t) == indep.exact_l2sq(q, t)


def test_uniqueness_is_lowest_query_index():
    # FeatureMatcher.cpp:53-64 walks queries in ascending order: the first claimant keeps the row
    rng = np.random.default_rng(5)
    t = (rng.standard_normal((30, 16)) * 5).astype(np.float32)
    q = np.repeat(t[[4, 9]], 3, axis=0) + rng.standard_normal((6, 16)).astype(np.float32) * 0.01
    out, n = orc.match_pair(q, t)
    assert list(out) == [4, -1, -1, 9, -1, -1] and n == 2


def test_ragged_and_threads_agree():
    ims = synth.descriptor_set("sift", 5, [40, 0, 33, 1, 64], n_world=100, seed=11)
    pairs = orc.all_pairs(5)
    o1, c1 = orc.match_grid(ims, pairs, threads=1)
    o4, c4 = orc.match_grid(ims, pairs, threads=4)
    assert np.array_equal(o1, o4) and np.array_equal(c1, c4)
    for p, (a, b) in enumerate(pairs):
        if ims[a].shape[0]:
            assert np.array_equal(o1[p, :ims[a].shape[0]], indep.numpy_match_pair(ims[a], ims[b]))
        assert (o1[p, ims[a].shape[0]:] == -1).all()


def test_oracle_power_of_two_scale_invariance():
    ims = synth.descriptor_set("superpoint", 2, [120, 150], n_world=300, seed=31)
    base, _ = orc.match_pair(ims[0], ims[1])
    for k in (-40, 9, 33):
        f = np.float32(2.0 ** k)
        assert np.array_equal(orc.match_pair(ims[0] * f, ims[1] * f)[0], base)

"""CPU suite: the error model behind the certified re-rank, checked on an
emulation of the coarse pass: inputs scaled by the global power of two and rounded to fp16,
products accumulated in fp32 on top of the fp32 half-norm, compared with the exact real value."""
import numpy as np
import pytest

from reconstructor_amd import synth


def _emulate(q, t):
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    maxabs = max(np.abs(q64).max(), np.abs(t64).max())
    maxn2 = max((q64 ** 2).sum(1).max(), (t64 ** 2).sum(1).max())
    _, ex = np.frexp(maxabs)
    s = np.ldexp(1.0, 14 - int(ex))
    mn, en = np.frexp(maxn2 * (1 + 1e-12))
    maxn2q = np.ldexp(np.ceil(mn * 32.0) / 32.0, int(en))
    bias = 0.5625 * s * s * maxn2q + 1.0
    nmax = np.sqrt(maxn2q)
    DP = max(32, 1 << int(np.ceil(np.log2(q.shape[1]))))
    a = (s * t64).astype(np.float16)                       # train operand
    b = (-s * q64).astype(np.float16)                      # negated query operand
    hn = (0.5 * s * s * (t64 ** 2).sum(1) + bias).astype(np.float32)
    # fp32 accumulation (order differs from the MFMA's; the bound must hold for any order)
    acc = hn[None, :] + (b.astype(np.float32) @ a.astype(np.float32).T)
    acc = acc.astype(np.float32).astype(np.float64)
    exact = 0.5 * s * s * (t64 ** 2).sum(1)[None, :] + bias - s * s * (q64 @ t64.T)
    u = 2.0 ** -11
    nq = np.sqrt((q64 ** 2).sum(1)) * (1 + 1e-12)
    hn_max = 0.5 * s * s * nmax * nmax + bias
    eps = ((2 * u + u * u) * s * s * nq * nmax + 2.0 ** -14 * np.sqrt(DP) * s * (nq + nmax) + 1e-9 +
           (DP + 8) * 2.0 ** -23 * (hn_max + s * s * nq * nmax) + 6.0e-8 * hn_max)
    return acc, exact, eps, s, bias


@pytest.mark.parametrize("kind,K,scale", [("superpoint", 300, 1.0), ("sift", 200, 1.0), ("orb", 250, 1.0),
                                          ("superpoint", 200, 2.0 ** -30), ("sift", 150, 2.0 ** 40)])
def test_coarse_error_is_inside_the_bound(kind, K, scale):
    ims = synth.descriptor_set(kind, 2, K, n_world=3 * K, seed=4)
    q, t = (ims[0] * np.float32(scale)).astype(np.float32), (ims[1] * np.float32(scale)).astype(np.float32)
    acc, exact, eps, s, bias = _emulate(q, t)
    err = np.abs(acc - exact)
    assert (err <= eps[:, None]).all(), (err.max(), eps.min())
    assert err.max() <= 0.6 * eps.max()                    # and not by a hair
    assert (acc > 0).all()                                 # positive floats order like unsigned integers
    # in squared-distance units the bound is the small number quoted in DESIGN.md for unit-norm data
    if kind == "superpoint":
        assert 2.0 * eps.max() / (s * s) / scale ** 2 < 3e-3


def test_mixed_magnitudes():
    rng = np.random.default_rng(2)
    q = (rng.standard_normal((64, 64)) * 10.0 ** rng.integers(-6, 1, (64, 1))).astype(np.float32)
    t = (rng.standard_normal((80, 64)) * 10.0 ** rng.integers(-6, 1, (80, 1))).astype(np.float32)
    acc, exact, eps, s, bias = _emulate(q, t)
    assert (np.abs(acc - exact) <= eps[:, None]).all() and (acc > 0).all()


@pytest.mark.parametrize("kind,K", [("superpoint", 400), ("sift", 300)])
def test_filter_decisions_agree_with_exact_knn(kind, K):
    """k_filter's two certificates (match.hip), emulated: rows it certifies as 'no match' or as
    'match = coarse best' must agree with the exact 2-NN + ratio test of the oracle."""
    from oracle import orc
    ims = synth.descriptor_set(kind, 2, K, n_world=2 * K, seed=9)
    q, t = ims[0], ims[1]
    acc, exact, eps, s, bias = _emulate(q, t)
    K2 = t.shape[0]
    bits = max(1, int(np.ceil(np.log2(K2))))
    mask = np.uint32((1 << bits) - 1)
    packed = (acc.astype(np.float32).view(np.uint32) & ~mask) | np.arange(K2, dtype=np.uint32)[None, :]
    packed.sort(axis=1)
    c0, c1 = packed[:, 0], packed[:, 1]
    f = lambda u: u.astype(np.uint32).view(np.float32).astype(np.float64)
    nq2 = (q.astype(np.float64) ** 2).sum(1)
    d2 = lambda a_: nq2 + (2.0 / (s * s)) * (a_ - bias)
    lo0, hi0 = f(c0 & ~mask), f((c0 & ~mask) + mask + np.uint32(1))
    lo1, hi1 = f(c1 & ~mask), f((c1 & ~mask) + mask + np.uint32(1))
    lb0, ub1 = np.maximum(d2(lo0 - eps), 0.0), d2(hi1 + eps)
    ub0, lbnc = d2(hi0 + eps), np.maximum(d2(lo1 - eps), 0.0)
    rp = lambda a_, b_: np.sqrt(a_.astype(np.float32)) < np.float32(0.7) * np.sqrt(b_.astype(np.float32))
    cert_fail = ~rp(lb0, ub1)
    cert_pass = ~cert_fail & (ub0 < lbnc) & rp(ub0, lbnc)
    idx, dd = orc.knn2(q, t)
    truth = np.where(rp(dd[:, 0], dd[:, 1]), idx[:, 0], -1)
    assert (truth[cert_fail] == -1).all()
    assert (truth[cert_pass] == (c0 & mask).astype(np.int64)[cert_pass]).all()
    # the certificates decide nearly everything on descriptor-like data
    assert (cert_fail | cert_pass).mean() > 0.98 and cert_pass.sum() > 0

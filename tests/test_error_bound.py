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

"""GPU parity suite for the matcher: HIP path (through the C ABI) vs the CPU oracle,
bit-exact on indices.  Run on the MI355X box with -m gpu."""
import os

import numpy as np
import pytest

from oracle import orc
from reconstructor_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def matcher(gpu_ctx):
    from reconstructor_amd.matcher import HipL2Matcher
    return HipL2Matcher(ctx=gpu_ctx)


def _cases(npz):
    z = np.load(os.path.join(G, npz))
    names = sorted({k.split("/")[0] for k in z.files})
    return [(n, z[n + "/q"], z[n + "/t"], z[n + "/expect"]) for n in names]


@pytest.mark.parametrize("name,q,t,expect", _cases("match_kat.npz"), ids=lambda v: v if isinstance(v, str) else "")
def test_kat(matcher, name, q, t, expect):
    assert np.array_equal(matcher.match_pair(q, t), expect)


@pytest.mark.parametrize("name,q,t,expect", _cases("match_seeded.npz"), ids=lambda v: v if isinstance(v, str) else "")
def test_seeded(matcher, name, q, t, expect):
    assert np.array_equal(matcher.match_pair(q, t), expect)


def test_plugin_signature_fills_map(matcher):
    """FeatureMatcher::matchFeatures(features1, features2, matches, shape1, shape2)."""
    z = np.load(os.path.join(G, "match_seeded.npz"))
    q, t, exp = z["sift/q"], z["sift/t"], z["sift/expect"]
    matches = {}
    matcher.match_features(q, t, matches, (336, 512), (336, 512))
    assert matches == {int(i): int(exp[i]) for i in np.nonzero(exp >= 0)[0]}


def test_grid8_golden(matcher):
    z = np.load(os.path.join(G, "match_grid8.npz"))
    ks = list(z["ks"])
    ims = synth.descriptor_set("superpoint", 8, ks, n_world=int(z["n_world"]), seed=int(z["seed"]))
    matcher.clear()
    for i, im in enumerate(ims):
        matcher.upload(i, im)
    out, counts = matcher.match_grid(z["pairs"], max(ks))
    assert np.array_equal(out, z["expect"])
    assert np.array_equal(counts, z["counts"])
    matcher.clear()


@pytest.mark.parametrize("kind,K1,K2", [("superpoint", 700, 1000), ("superpoint", 2048, 2048),
                                        ("sift", 900, 640), ("orb", 500, 777)])
def test_pair_vs_oracle(matcher, kind, K1, K2):
    ims = synth.descriptor_set(kind, 2, [K1, K2], n_world=3000, seed=21)
    exp, n = orc.match_pair(ims[0], ims[1])
    got = matcher.match_pair(ims[0], ims[1])
    assert np.array_equal(got, exp)
    st = matcher.stats()
    assert st["used_mfma_path"] == 1
    assert st["pair_distances"] == K1 * K2
    print(kind, K1, K2, "matches", n, "fallback rows", st["rows_exact_fallback"], "eps_d2", st["err_bound_d2"])


def test_ragged_grid_vs_oracle(matcher):
    ks = [300, 0, 257, 1, 512, 2, 640]
    ims = synth.descriptor_set("superpoint", len(ks), ks, n_world=900, seed=5)
    matcher.clear()
    for i, im in enumerate(ims):
        matcher.upload(i, im)
    pairs = orc.all_pairs(len(ks))
    # both orientations: the grid API takes any (query, train) list
    pairs = np.concatenate([pairs, pairs[:, ::-1]]).astype(np.int32)
    exp, ec = orc.match_grid(ims, pairs, threads=4)
    out, counts = matcher.match_grid(pairs, max(ks))
    assert np.array_equal(out, exp)
    assert np.array_equal(counts, ec)
    matcher.clear()


def test_adversarial_near_duplicates(matcher):
    """Many train rows within the coarse error bound of each other: the certified re-rank
    must hand these rows to the exact kernel and still be bit-exact."""
    rng = np.random.default_rng(8)
    base = rng.standard_normal((40, 256)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    t = np.repeat(base, 16, axis=0) + (rng.standard_normal((640, 256)) * 1e-5).astype(np.float32)
    q = base + (rng.standard_normal((40, 256)) * 1e-5).astype(np.float32)
    exp, _ = orc.match_pair(q, t)
    assert np.array_equal(matcher.match_pair(q, t), exp)
    # the coarse pair cannot decide any of these rows (sixteen train rows within the error band of each other: neither certificate of
    # k_filter holds), so every query row must have gone through the exact fp64 re-rank -- the tier this test is about did run
    st = matcher.stats()
    assert st["rows_total"] == 40 and st["rows_reranked"] == 40, st


def test_unsupported_dim_takes_exact_kernel(matcher):
    rng = np.random.default_rng(9)
    q = rng.standard_normal((130, 300)).astype(np.float32)   # D > 256: no MFMA path
    t = np.concatenate([q[:50] + 0.01, rng.standard_normal((100, 300)).astype(np.float32)]).astype(np.float32)
    exp, _ = orc.match_pair(q, t)
    assert np.array_equal(matcher.match_pair(q, t), exp)
    assert matcher.stats()["used_mfma_path"] == 0
    q = rng.standard_normal((65, 7)).astype(np.float32)       # D not a multiple of 4
    t = rng.standard_normal((90, 7)).astype(np.float32)
    exp, _ = orc.match_pair(q, t)
    assert np.array_equal(matcher.match_pair(q, t), exp)


def test_full_size_properties(matcher):
    """BASELINE cfg-2 shapes (2048 x 2048 x 256): size-independent properties."""
    ims = synth.descriptor_set("superpoint", 3, 2048, seed=1234)
    matcher.clear()
    for i, im in enumerate(ims):
        matcher.upload(i, im)
    pairs = np.array([[0, 0], [0, 1], [1, 0], [1, 2]], np.int32)
    out, counts = matcher.match_grid(pairs, 2048)
    # self pair: every row's nearest neighbour is itself at distance 0 -> identity map
    assert np.array_equal(out[0], np.arange(2048))
    for p in range(4):
        m = out[p][out[p] >= 0]
        assert len(m) == counts[p] and len(np.unique(m)) == len(m)      # uniqueness
    # a sample of rows against the oracle's exact 2-NN
    idx, d2 = orc.knn2(ims[0][:64], ims[1])
    for i in range(64):
        passes = np.sqrt(np.float32(d2[i, 0])) < np.float32(0.7) * np.sqrt(np.float32(d2[i, 1]))
        assert out[1][i] in ((idx[i, 0], -1) if passes else (-1,))
    matcher.clear()


def test_cfg3_shape_vs_oracle(matcher):
    """BASELINE cfg-3 shape (4096 keypoints per image, 12 index bits in the packed top-2):
    one full pair against the oracle, bit-exact, and the certified fast path must carry it."""
    ims = synth.descriptor_set("superpoint", 2, 4096, seed=77)
    exp, n = orc.match_pair(ims[0], ims[1])
    got = matcher.match_pair(ims[0], ims[1])
    assert np.array_equal(got, exp)
    st = matcher.stats()
    assert st["used_mfma_path"] == 1 and st["rows_exact_fallback"] < 0.01 * 4096
    print("cfg3 pair: matches", n, "reranked", st["rows_reranked"], "fallback", st["rows_exact_fallback"])


@pytest.mark.parametrize("seed", range(10))
def test_random_grids_vs_oracle(matcher, seed):
    """Randomised differential test: ragged image sizes (including 0, 1, 2 and non-multiples of
    the 64-row tile / 512-row work item), descriptor lengths on and off the MFMA path, pair lists
    with repeats, both orientations and self pairs, integer-valued data with exact ties."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.choice([3, 8, 32, 40, 64, 100, 128, 256, 260]))
    n = int(rng.integers(2, 8))
    ks = [int(rng.choice([0, 1, 2, 63, 64, 65, 200, 511, 512, 513, 700])) for _ in range(n)]
    kind = int(rng.integers(0, 3))
    ims = []
    for k in ks:
        if kind == 0:
            a = rng.standard_normal((k, D)).astype(np.float32)
        elif kind == 1:
            a = rng.integers(0, 4, (k, D)).astype(np.float32)            # many exact ties
        else:
            a = (rng.standard_normal((k, D)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        ims.append(np.ascontiguousarray(a))
    if kind == 0 and n >= 2 and ks[0] and ks[1]:                            # plant true matches
        m = min(ks[0], ks[1])
        ims[1][:m] = ims[0][:m] + (rng.standard_normal((m, D)) * 0.05).astype(np.float32)
    P = int(rng.integers(1, 25))
    pairs = rng.integers(0, n, (P, 2)).astype(np.int32)
    if rng.random() < 0.5:
        pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]               # runs that share the query image
    matcher.clear()
    for i, im in enumerate(ims):
        matcher.upload(i, im if im.shape[0] else np.zeros((0, D), np.float32))
    kmax = max(1, max(ks))
    out, counts = matcher.match_grid(pairs, kmax)
    exp, ec = orc.match_grid([im if im.shape[0] else np.zeros((0, D), np.float32) for im in ims], pairs, threads=2)
    assert np.array_equal(out, exp), (seed, D, ks, pairs.tolist())
    assert np.array_equal(counts, ec)
    matcher.clear()


@pytest.mark.parametrize("kind", ["superpoint", "sift"])
def test_power_of_two_scale_invariance(matcher, kind):
    """Scaling every descriptor by 2^k commutes with every operation of the canonical matcher
    (no under/overflow here), so the match table must not change: exercises the global
    power-of-two scale, the bias and the error-bound constants end to end."""
    ims = synth.descriptor_set(kind, 2, [600, 700], n_world=1500, seed=31)
    base = matcher.match_pair(ims[0], ims[1])
    assert np.array_equal(base, orc.match_pair(ims[0], ims[1])[0])
    for k in (-40, -7, 9, 33):
        f = np.float32(2.0 ** k)
        assert np.array_equal(matcher.match_pair(ims[0] * f, ims[1] * f), base), k


def test_more_than_8192_train_rows_stay_on_the_mfma_path(matcher):
    """K2 > 8192: the row index takes 14..16 of the candidate's 32 bits (a coarser value: more rows re-ranked, same
    answers) and the owner table of the uniqueness step no longer fits LDS (global owner table)."""
    rng = np.random.default_rng(17)
    t = rng.standard_normal((9000, 32)).astype(np.float32)
    q = np.concatenate([t[rng.choice(9000, 200, replace=False)] + 0.05 * rng.standard_normal((200, 32)).astype(np.float32),
                        rng.standard_normal((150, 32)).astype(np.float32)])
    q[5] = q[4]                                            # two queries claiming one train row
    got = matcher.match_pair(q, t)
    exp, cnt = orc.match_pair(q, t)
    assert np.array_equal(got, exp) and cnt > 100
    matcher.clear()
    matcher.upload(0, q); matcher.upload(1, t)
    out, counts = matcher.match_grid(np.array([[0, 1], [1, 0]], np.int32), 9000)
    exp2, c2 = orc.match_grid([q, t], np.array([[0, 1], [1, 0]]), threads=4)
    assert np.array_equal(out, exp2) and np.array_equal(counts, c2)
    assert matcher.stats()["used_mfma_path"] == 1
    matcher.clear()


@pytest.mark.parametrize("kind,K", [("sift", 12000), ("superpoint", 20000)])
def test_large_images_match_the_oracle(matcher, kind, K):
    """Well past 8192 keypoints per image (a full-resolution SIFT run; the reference resizes to 512 px and never gets
    there): 14 / 15 index bits, most rows still certified by the coarse pass alone."""
    ims = synth.descriptor_set(kind, 2, K, n_world=2 * K, seed=29)
    matcher.clear()
    matcher.upload(0, ims[0]); matcher.upload(1, ims[1])
    out, counts = matcher.match_grid(np.array([[0, 1], [1, 0]], np.int32), K)
    st = matcher.stats()
    exp, c = orc.match_grid(ims, np.array([[0, 1], [1, 0]]), threads=8)
    assert np.array_equal(out, exp) and np.array_equal(counts, c) and c.min() > K // 20
    assert st["used_mfma_path"] == 1 and st["rows_exact_fallback"] < 0.2 * st["rows_total"], st
    matcher.clear()


def test_more_than_65536_rows_take_the_generic_path(matcher):
    """Beyond 16 index bits the exact kernel answers alone."""
    rng = np.random.default_rng(5)
    t = rng.standard_normal((66000, 8)).astype(np.float32)
    q = np.concatenate([t[rng.choice(66000, 40, replace=False)] + 0.01 * rng.standard_normal((40, 8)).astype(np.float32),
                        rng.standard_normal((24, 8)).astype(np.float32)])
    got = matcher.match_pair(q, t)
    exp, cnt = orc.match_pair(q, t)
    assert np.array_equal(got, exp) and cnt >= 30
    assert matcher.stats()["used_mfma_path"] == 0
    matcher.clear()


def test_exactly_8192_rows_on_the_mfma_path(matcher):
    """K = 8192: the largest image with a 13-bit row index and an LDS owner table."""
    ims = synth.descriptor_set("orb", 2, 8192, n_world=20000, seed=23)
    exp, cnt = orc.match_pair(ims[0], ims[1])
    assert np.array_equal(matcher.match_pair(ims[0], ims[1]), exp) and cnt > 500
    assert matcher.stats()["used_mfma_path"] == 1


def test_null_pair_list_is_the_canonical_grid(matcher):
    """pairs == NULL: every i < j over the resident ids in ascending order (the reference's pair loop)."""
    ims = synth.descriptor_set("superpoint", 5, [90, 130, 70, 110, 60], n_world=300, seed=3)
    matcher.clear()
    ids = [7, 2, 11, 5, 3]
    for i, im in zip(ids, ims):
        matcher.upload(i, im)
    out, counts = matcher.match_all_pairs(5, 130)
    order = sorted(range(5), key=lambda k: ids[k])
    pairs = [(order[a], order[b]) for a in range(5) for b in range(a + 1, 5)]
    exp, ec = orc.match_grid(ims, np.array(pairs), threads=2)
    assert np.array_equal(out[:, :exp.shape[1]], exp) and np.array_equal(counts, ec)
    with pytest.raises(Exception):
        matcher.ctx.check(matcher.ctx.lib.rcn_match_grid(matcher.ctx.h, None, 7, 0.7, out.ctypes.data, 130, counts.ctypes.data))
    matcher.clear()


def test_cfg2_full_grid_equals_the_oracle(matcher):
    """BASELINE.json configs[1] in full: all 4950 pairs of 100 x 2048 x 256, every pair's row hash and match count
    against the CPU oracle's (tests/golden/match_cfg2_full.npz, written by tests/golden/make_cfg2_golden.py)."""
    import torch
    from reconstructor_amd import tablehash
    g = np.load(os.path.join(G, "match_cfg2_full.npz"))
    n, K, seed = int(g["n_images"]), int(g["K"]), int(g["seed"])
    pool = synth.world_pool("superpoint", 4 * K, seed=seed)
    block = torch.from_numpy(np.stack([synth.image_descriptors("superpoint", i, K, pool, seed=seed) for i in range(n)])).cuda()
    P = len(g["pairs"])
    out = torch.empty((P, K), dtype=torch.int32, device="cuda")
    cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    matcher.clear()
    matcher.upload_batch_device(0, n, block.data_ptr(), K, 256)
    matcher.match_grid_device(g["pairs"], out.data_ptr(), K, cnt.data_ptr())
    matcher.ctx.check(matcher.ctx.lib.rcn_synchronize(matcher.ctx.h))
    h, c = tablehash.row_hashes(out.cpu().numpy(), K)
    assert np.array_equal(c, g["counts"]) and np.array_equal(cnt.cpu().numpy(), g["counts"])
    bad = np.nonzero(h != g["hashes"])[0]
    assert len(bad) == 0, ("pairs whose table differs from the oracle's", g["pairs"][bad][:8])
    assert int(c.sum()) == int(g["matches_found"])
    matcher.clear()


def test_out_of_scale_rows_are_set_aside_not_followed(matcher):
    """The fp16 scale of the coarse pass is one power of two for every resident image.  Rows far out of scale -- here one image
    2^30 times larger than the rest, as query and as train image, plus three stray huge rows inside an ordinary image -- are
    set aside as BIG rows (fix_scale's histogram of row norms): exact kernel as queries, never coarse candidates, a norm bound
    in everybody else's certificates.  The grid stays equal to the oracle; the pairs that do not involve the large image equal
    the run without it; and the ordinary rows keep their certificates instead of all falling back (what following the largest
    row with the scale would cost: every fp16 copy underflows)."""
    n, K = 40, 120                                         # 4800 rows, 124 of them out of scale (the budget is one row in eight)
    ims = synth.descriptor_set("sift", n, K, n_world=400, seed=23)
    pairs = orc.all_pairs(n).astype(np.int32)
    matcher.clear()
    for i, im in enumerate(ims):
        matcher.upload(i, im)
    base, _ = matcher.match_grid(pairs, K)
    st0 = matcher.stats()
    big = [im.copy() for im in ims]
    big[17] = big[17] * np.float32(2.0 ** 30)              # a whole image out of scale: query in (17, j), train in (i, 17)
    big[3][[5, 60, 119]] *= np.float32(2.0 ** 28)          # three stray rows of an ordinary image
    big[3][61] = big[3][60]                                # ... one of them duplicated: a BIG query whose nearest neighbour is BIG
    matcher.clear()
    for i, im in enumerate(big):
        matcher.upload(i, im)
    out, counts = matcher.match_grid(pairs, K)
    st1 = matcher.stats()
    exp, ec = orc.match_grid(big, pairs, threads=0)
    assert np.array_equal(out, exp) and np.array_equal(counts, ec)
    keep = ((pairs != 17) & (pairs != 3)).all(1)
    assert np.array_equal(out[keep], base[keep])
    # to the exact kernel: BIG query rows (image 17's rows in its 22 pairs as query, image 3's four rows in its 36 pairs) and
    # every query row of the 17 pairs whose TRAIN image is image 17 (no ordinary train row to propose)
    assert st1["rows_exact_fallback"] <= st0["rows_exact_fallback"] + (n - 1) * K + 4 * (n - 1 - 3) + 64
    assert st1["rows_exact_fallback"] >= (n - 1) * K
    print("fallback rows: uniform %d, with out-of-scale rows %d of %d" % (st0["rows_exact_fallback"], st1["rows_exact_fallback"], st1["rows_total"]))
    matcher.clear()


def test_host_batch_ingest_equals_one_upload_per_image(matcher):
    """rcn_desc_upload_batch: ragged [n][K_i][D] host images in one call (one block, one stats launch, one
    synchronisation) -- the tables equal the one-upload-per-image path's and the oracle's; a second batch of the same
    shape reuses the block; an empty image in the middle and an image of one row are part of it."""
    ks = [300, 0, 513, 1, 256, 777]
    ims = synth.descriptor_set("sift", len(ks), [max(k, 1) for k in ks], n_world=1200, seed=91)
    ims = [im[:k] for im, k in zip(ims, ks)]
    pairs = [(0, 2), (0, 4), (2, 5), (4, 5), (3, 5), (5, 0), (1, 2), (2, 1)]
    exp, ec = orc.match_grid(ims, pairs, threads=2)
    matcher.clear()
    matcher.upload_batch(10, ims)
    shifted = [(a + 10, b + 10) for a, b in pairs]
    out, counts = matcher.match_grid(shifted, max(ks))
    assert np.array_equal(out, exp) and np.array_equal(counts, ec)
    # same shape again, other contents: allocations reused, results follow the new rows
    ims2 = [im[::-1].copy() for im in ims]
    exp2, ec2 = orc.match_grid(ims2, pairs, threads=2)
    matcher.upload_batch(10, ims2)
    out, counts = matcher.match_grid(shifted, max(ks))
    assert np.array_equal(out, exp2) and np.array_equal(counts, ec2)
    # one image replaced on its own, one removed: the others stay
    matcher.upload(12, ims[2])
    matcher.remove(15)
    assert matcher.ctx.lib.rcn_desc_count(matcher.ctx.h) == len(ks) - 1
    ims3 = list(ims2)
    ims3[2] = ims[2]
    exp3, ec3 = orc.match_grid(ims3, [(0, 2), (0, 4), (2, 4)], threads=2)
    out, counts = matcher.match_grid([(10, 12), (10, 14), (12, 14)], max(ks))
    assert np.array_equal(out, exp3) and np.array_equal(counts, ec3)
    matcher.clear()
    assert matcher.ctx.lib.rcn_desc_count(matcher.ctx.h) == 0


def test_remove_frees_the_descriptor_length(matcher):
    """rcn_desc_remove of the last image lets the ctx take another D (what the plugin's cache does on a shared ctx
    instead of rcn_desc_clear, which would wipe everybody's descriptors)."""
    from reconstructor_amd import _lib
    matcher.clear()
    a = synth.descriptor_set("sift", 2, 200, n_world=400, seed=5)
    b = synth.descriptor_set("orb", 2, 200, n_world=400, seed=6)
    matcher.upload(1, a[0])
    matcher.upload(2, a[1])
    with pytest.raises(_lib.RcnError):
        matcher.upload(3, b[0])                 # D = 32 beside D = 128
    matcher.remove(1)
    matcher.remove(2)
    matcher.remove(2)                           # not resident any more: a no-op
    matcher.upload(3, b[0])
    matcher.upload(4, b[1])
    out, counts = matcher.match_grid([(3, 4)], 200)
    exp, ec = orc.match_grid(b, [(0, 1)], threads=1)
    assert np.array_equal(out, exp) and np.array_equal(counts, ec)
    matcher.clear()

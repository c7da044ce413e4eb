"""GPU suite: rcn_fmat_filter* (fmat.hip) against the oracle -- inlier masks, counts and iteration
counts, bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import orc_fmat as of
from reconstructor_amd import _lib, fmat, synth_fmat

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fmat_small.npz")


def test_golden_fixture(gpu_ctx):
    g = np.load(GOLD)
    mask, counts, iters = fmat.filter_grid(gpu_ctx, g["pair_off"], g["xy1"], g["xy2"])
    assert (counts == g["counts"]).all() and (iters == g["iterations"]).all() and (mask == g["mask"]).all()


@pytest.mark.parametrize("frac,seed", [(0.1, 1), (0.3, 2), (0.6, 3)])
def test_seeded_grids_match_oracle(gpu_ctx, frac, seed):
    sizes = [0, 6, 7, 14, 15, 16] + list(np.random.default_rng(seed).integers(15, 900, 120))
    off, a, b = synth_fmat.grid(sizes, frac, seed=seed)
    want = of.filter_grid(off, a, b, threads=8)
    got = fmat.filter_grid(gpu_ctx, off, a, b)
    assert (got[1] == want[1]).all() and (got[2] == want[2]).all() and (got[0] == want[0]).all()
    assert (want[1][6:] >= 7).all()


def test_single_pair_entry_and_sizes_beyond_the_lds_copy(gpu_ctx):
    for n, seed in [(7, 1), (20, 2), (2048, 3), (2049, 4), (5000, 5)]:
        a, b, _ = synth_fmat.two_view(n, 0.3, seed=seed)
        m0, c0, _ = of.filter_pair(a, b)
        m1, c1 = fmat.estimate_fundamental_inliers(gpu_ctx, a, b)
        assert c0 == c1 and (m0 == m1).all()
    m, c = fmat.estimate_fundamental_inliers(gpu_ctx, a[:4], b[:4])
    assert c == -2 and m.all()
    # the winning matrix: rank 2, F(3,3) = 1, and it reproduces the mask
    a, b, _ = synth_fmat.two_view(200, 0.3, seed=9)
    m, c, F = fmat.estimate_fundamental_inliers(gpu_ctx, a, b, with_matrix=True)
    assert F[2, 2] == 1.0 and abs(np.linalg.det(F)) < 1e-12 * np.abs(F).max() ** 3 + 1e-18
    err = np.array([of.epi_error(F, a[i], b[i]) for i in range(200)])
    assert ((err <= 9.0) == m).all() and m.sum() == c
    same = np.tile([[100, 100]], (30, 1)).astype(np.int32)
    m, c = fmat.estimate_fundamental_inliers(gpu_ctx, same, same)          # the draw gives up: no model
    assert c == -1 and not m.any()


def test_errors(gpu_ctx):
    a, b, _ = synth_fmat.two_view(20, 0.2, seed=1)
    with pytest.raises(_lib.RcnError):
        fmat.filter_grid(gpu_ctx, [0, 30, 20], np.r_[a, a], np.r_[b, b])     # offsets not monotone
    rc = gpu_ctx.lib.rcn_fmat_filter(gpu_ctx.h, None, None, 5, None, None, None)
    assert rc == -1
    mask, counts, iters = fmat.filter_grid(gpu_ctx, [0], a[:0], b[:0])       # empty grid
    assert len(counts) == 0


def test_matcher_output_through_the_filter(gpu_ctx):
    """The slice of the pair loop this repo covers (SequentialReconstructor.cpp:202-269): match every
    pair, then keep the epipolar inliers -- GPU path against the oracles' composition."""
    from oracle import orc
    from reconstructor_amd import synth
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    K, n_img = 220, 4
    ims = synth.descriptor_set("superpoint", n_img, K, n_world=500, seed=8)
    rng = np.random.default_rng(8)
    coords = [rng.integers(0, 500, (K, 2)).astype(np.int32) for _ in range(n_img)]
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    for i, im in enumerate(ims):
        m.upload(i, im)
    pairs = all_pairs(n_img)
    table, _ = m.match_grid(pairs, K)
    m.clear()
    exp_table, _ = orc.match_grid(ims, pairs, threads=2)
    assert np.array_equal(table, exp_table)
    out, counts = fmat.apply_geometric_filter(gpu_ctx, coords, pairs, table)
    off, a, b = fmat.matches_to_csr(coords, pairs, exp_table)
    mask, ocounts, _ = of.filter_grid(off, a, b)
    assert (counts == ocounts).all()
    for p in range(len(pairs)):
        q = np.flatnonzero(exp_table[p] >= 0)
        keep = mask[off[p]:off[p + 1]] if len(q) >= 7 else np.ones(len(q), bool)
        want = np.full(K, -1, np.int32)
        want[q[keep]] = exp_table[p, q[keep]]
        assert np.array_equal(out[p], want)


def test_full_size_grid_properties_and_device_entry(gpu_ctx):
    """cfg-2-sized grid (4950 pairs, ~500 matches each): counts never exceed the pair's size, the
    mask sums equal the counts, outliers are rejected; a 600-pair prefix is compared outright."""
    import torch
    sizes = np.random.default_rng(0).integers(300, 700, 4950)
    off, a, b = synth_fmat.grid(sizes[:600], 0.3, seed=11)
    want = of.filter_grid(off, a, b, threads=16)
    d = {k: torch.from_numpy(v).cuda() for k, v in (("off", off), ("a", a), ("b", b))}
    mask_d = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda")
    cnt_d = torch.zeros(600, dtype=torch.int32, device="cuda")
    it_d = torch.zeros(600, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.check(gpu_ctx.lib.rcn_fmat_filter_grid_device(gpu_ctx.h, 600, d["off"].data_ptr(), d["a"].data_ptr(), d["b"].data_ptr(),
                                                          mask_d.data_ptr(), cnt_d.data_ptr(), it_d.data_ptr(), None))
    gpu_ctx.check(gpu_ctx.lib.rcn_synchronize(gpu_ctx.h))
    assert (cnt_d.cpu().numpy() == want[1]).all() and (it_d.cpu().numpy() == want[2]).all()
    assert (mask_d.cpu().numpy().astype(bool) == want[0]).all()
    sums = np.add.reduceat(want[0].astype(np.int64), off[:-1])
    assert (sums == want[1]).all() and (want[1] <= np.diff(off)).all()


def test_device_table_filter_equals_host_composition(gpu_ctx):
    """rcn_match_grid_device -> rcn_match_table_filter_device (everything stays in HBM) against the host
    composition of the same steps, itself checked against the oracles above; ragged image sizes, a pair
    with fewer than 7 matches, and a pair whose matches admit no model."""
    import torch
    from reconstructor_amd import synth
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    Ks = [260, 300, 180, 40, 230]
    ims = synth.descriptor_set("superpoint", 5, Ks, n_world=420, seed=12)
    rng = np.random.default_rng(12)
    coords = [rng.integers(0, 500, (k, 2)).astype(np.int32) for k in Ks]
    coords[4][:] = 77                                       # every sample of pairs (i, 4) is collinear on the train side: no model
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    for i, im in enumerate(ims):
        m.upload(i, im)
        m.upload_coords(i, coords[i])
    pairs = all_pairs(5)
    stride = max(Ks)
    table, counts = m.match_grid(pairs, stride)
    want, status = fmat.apply_geometric_filter(gpu_ctx, coords, pairs, table)
    t_d = torch.full((len(pairs), stride), -7, dtype=torch.int32, device="cuda")
    c_d = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    s_d = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m.match_grid_device(pairs, t_d.data_ptr(), stride, c_d.data_ptr())
    m.filter_table_device(pairs, t_d.data_ptr(), stride, c_d.data_ptr(), s_d.data_ptr())
    gpu_ctx.check(gpu_ctx.lib.rcn_synchronize(gpu_ctx.h))
    got = t_d.cpu().numpy()
    assert np.array_equal(got, want) and np.array_equal(s_d.cpu().numpy(), status)
    assert np.array_equal(c_d.cpu().numpy(), (want >= 0).sum(1))
    assert (status == -1).any() and (status >= 7).any() and ((counts >= 7) | (status == -2)).all()
    with pytest.raises(_lib.RcnError) as e:                 # coordinates of an image missing
        m.filter_table_device(np.array([[0, 9]], np.int32), t_d.data_ptr(), stride, c_d.data_ptr())
    assert e.value.code == -5
    m.clear()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_redraw_heavy_and_degenerate_inputs(gpu_ctx, seed):
    """Inputs that push the sampler through its rejection paths: points on a coarse integer lattice (most
    samples have a collinear last point and are redrawn), heavy duplication (few distinct points),
    large coordinates, and pairs that are exact copies of each other."""
    rng = np.random.default_rng(seed)
    off, a, b = [0], [], []

    def add(x, y):
        a.append(np.asarray(x, np.int32)); b.append(np.asarray(y, np.int32)); off.append(off[-1] + len(x))
    for n in (15, 40, 90, 300):
        lat1 = rng.integers(0, 6, (n, 2)) * 40                       # 6 x 6 lattice
        lat2 = lat1 + rng.integers(-2, 3, (n, 2))
        add(lat1, lat2)
    for n in (16, 60, 200):
        base1, base2, _ = synth_fmat.two_view(8, 0.0, seed=seed * 10 + n)
        pick = rng.integers(0, 8, n)                                   # only 8 distinct correspondences
        add(base1[pick], base2[pick])
    x, y, _ = synth_fmat.two_view(120, 0.3, seed=seed + 50)
    add(x * 37 + 5000, y * 37 + 5000)                                   # coordinates up to ~25 000
    add(x, x)                                                          # identical images: F is not unique
    add(x[:25] * 0, y[:25])                                            # every point of image 1 at the origin
    off = np.asarray(off, np.int32); A = np.concatenate(a); B = np.concatenate(b)
    want = of.filter_grid(off, A, B, threads=8)
    got = fmat.filter_grid(gpu_ctx, off, A, B)
    assert (got[1] == want[1]).all() and (got[2] == want[2]).all() and (got[0] == want[0]).all()
    assert (want[1] == -1).any()                                       # some of these admit no model at all

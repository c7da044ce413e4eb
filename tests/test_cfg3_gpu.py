"""GPU suite: BASELINE.json configs[2] in full -- 1000 images x 4096 keypoints x 256-d, all 499 500 image pairs on
one GPU through the sharded-grid path (world size 1: the code every rank runs) -- checked against oracle outputs
for 64 sampled pairs outright (tests/golden/match_cfg3_sample.npz), for 512 more by row hash and count, 64 from every
residue of the pair number modulo 8 (match_cfg3_sample512.npz), for 5120 more stratified by pipeline chunk and residue
(match_cfg3_sample5k.npz; all written by tests/golden/make_cfg3_golden.py), and
through properties of the whole 8 GB match table."""
import os

import numpy as np
import pytest

from reconstructor_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "match_cfg3_sample.npz")
GOLD512 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "match_cfg3_sample512.npz")
GOLD5K = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "match_cfg3_sample5k.npz")


def test_cfg3_full_grid(gpu_ctx):
    import torch
    from reconstructor_amd import pairgrid
    g = np.load(GOLD)
    n, K, seed, D = int(g["n_images"]), int(g["K"]), int(g["seed"]), 256
    pool = synth.world_pool("superpoint", 4 * K, seed=seed)
    block = torch.empty((n, K, D), dtype=torch.float32, device="cuda")
    for i in range(n):
        block[i].copy_(torch.from_numpy(synth.image_descriptors("superpoint", i, K, pool, seed=seed)))
    P = n * (n - 1) // 2
    out = torch.empty((P, K), dtype=torch.int32, device="cuda")
    cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    sh = pairgrid.Shard(gpu_ctx, 0, 1, pairgrid.unique_id())
    try:
        gpu_ctx.check(gpu_ctx.lib.rcn_desc_clear(gpu_ctx.h))
        sh.reserve(n, K, D)
        sh.exchange(block.data_ptr())
        sh.match(0.7, out.data_ptr(), K, cnt.data_ptr())
        gpu_ctx.check(gpu_ctx.lib.rcn_synchronize(gpu_ctx.h))
        assert sh.info()["n_pairs"] == P
        # ---- the 64 sampled pairs, outright
        for (i, j), c, lo, hi in zip(g["pairs"], g["counts"], g["offsets"][:-1], g["offsets"][1:]):
            p = int(i) * (2 * n - int(i) - 1) // 2 + (int(j) - int(i) - 1)          # number of (i, j) in the canonical list
            row = out[p].cpu().numpy()
            q = np.nonzero(row >= 0)[0]
            qt = g["qt"][lo:hi].astype(np.int32)
            assert int(cnt[p].item()) == int(c) == len(q), (i, j)
            assert np.array_equal(q, qt[:, 0]) and np.array_equal(row[q], qt[:, 1]), (i, j)
        # ---- 512 more pairs, 64 from every rank's share of an 8-GPU deal (pair number mod 8), by row hash + count
        from reconstructor_amd import tablehash
        g5 = np.load(GOLD512)
        assert all(np.bincount(g5["pair_numbers"] % 8, minlength=8) == 64)
        rows = out[torch.from_numpy(g5["pair_numbers"]).cuda()].cpu().numpy()
        h, c = tablehash.row_hashes(rows, K)
        assert np.array_equal(c, g5["counts"]) and np.array_equal(cnt[torch.from_numpy(g5["pair_numbers"]).cuda()].cpu().numpy(), g5["counts"])
        assert np.array_equal(h, g5["hashes"]), g5["pairs"][h != g5["hashes"]][:8]
        # ---- round 5: 5120 more, 40 for every (sixteenth of the pair list = where the pipeline chunks fall) x (residue modulo 8)
        g6 = np.load(GOLD5K)
        nums = g6["pair_numbers"]
        assert len(nums) == 5120 and (np.bincount((nums * 16 // P) * 8 + nums % 8, minlength=128) == 40).all()
        rows = out[torch.from_numpy(nums).cuda()].cpu().numpy()
        h, c = tablehash.row_hashes(rows, K)
        assert np.array_equal(c, g6["counts"]) and np.array_equal(cnt[torch.from_numpy(nums).cuda()].cpu().numpy(), g6["counts"])
        assert np.array_equal(h, g6["hashes"]), g6["pairs"][h != g6["hashes"]][:8]
        # ---- the whole table: counts, index range, no train row claimed twice within a pair
        total = 0
        for a in range(0, P, 16384):
            t = out[a:a + 16384]
            assert int(t.min().item()) >= -1 and int(t.max().item()) < K
            valid = t >= 0
            assert torch.equal(valid.sum(1).to(torch.int32), cnt[a:a + 16384])
            s, _ = torch.sort(t, dim=1)
            dup = (s[:, 1:] == s[:, :-1]) & (s[:, 1:] >= 0)
            assert not bool(dup.any().item())
            total += int(valid.sum().item())
        assert total == int(cnt.to(torch.int64).sum().item()) and total > P        # ~1000 matches per pair on this set
        # ---- and the exact stages were really needed: some rows went through the fp64 re-rank / brute force
        from reconstructor_amd.matcher import HipL2Matcher
        st = HipL2Matcher(ctx=gpu_ctx).stats()
        assert st["used_mfma_path"] == 1 and st["rows_total"] == P * K and st["rows_reranked"] > 0
    finally:
        sh.close()


def test_a_2000_image_grid_fits_the_device(gpu_ctx):
    """Twice the images of cfg 3: 2000 x 4096 x 256, 1 999 000 pairs, 8.2e9 query rows.  Round 3 sized the candidate table
    and the two row lists for the whole grid (3 x 65 GB here, beside 33 GB of results: more than the device has); the
    pipeline chunks bound them at 0.5 GB each.  Checked through properties of the whole table, through the match the scene
    plants (row r of image i is a noisy copy of world point (37 i + r) mod 8192), and against the oracle for a few pairs."""
    import torch
    from oracle import orc
    from reconstructor_amd.matcher import HipL2Matcher
    n, K, D, W = 2000, 4096, 256, 8192
    gen = torch.Generator(device="cuda").manual_seed(77)
    world = torch.randn((W, D), generator=gen, device="cuda", dtype=torch.float32)
    world /= world.norm(dim=1, keepdim=True)
    block = torch.empty((n, K, D), dtype=torch.float32, device="cuda")
    ar = torch.arange(K, device="cuda")
    for i in range(n):
        rows = world[(37 * i + ar) % W] + 0.02 * torch.randn((K, D), generator=gen, device="cuda")
        block[i] = rows / rows.norm(dim=1, keepdim=True)
    P = n * (n - 1) // 2
    out = torch.empty((P, K), dtype=torch.int32, device="cuda")
    cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    free0, _ = torch.cuda.mem_get_info()
    m.upload_batch_device(0, n, block.data_ptr(), K, D)
    gpu_ctx.check(gpu_ctx.lib.rcn_match_grid_device(gpu_ctx.h, None, P, 0.7, out.data_ptr(), K, cnt.data_ptr()))
    gpu_ctx.check(gpu_ctx.lib.rcn_synchronize(gpu_ctx.h))
    free1, _ = torch.cuda.mem_get_info()
    st = m.stats()
    try:
        assert st["used_mfma_path"] == 1 and st["rows_total"] == P * K and st["chunks"] > 50
        # what the library itself holds for this grid: fp16 copies + norms of the images (4.4 GB) and the chunked workspace
        assert (free0 - free1) / 1e9 < 8.0, (free0 - free1) / 1e9
        total = 0
        for a in range(0, P, 16384):
            t = out[a:a + 16384]
            assert int(t.min().item()) >= -1 and int(t.max().item()) < K
            valid = t >= 0
            assert torch.equal(valid.sum(1).to(torch.int32), cnt[a:a + 16384])
            total += int(valid.sum().item())
        assert total > P
        # the planted correspondence on sampled pairs: query row r of image i and train row r' of image j show the same world
        # point when 37 i + r = 37 j + r' (mod 8192); nearly all of those rows must be matched, and to exactly that row
        rng = np.random.default_rng(5)
        for _ in range(12):
            i, j = sorted(rng.choice(n, 2, replace=False).tolist())
            p = i * (2 * n - i - 1) // 2 + (j - i - 1)
            row = out[p].cpu().numpy()
            r = np.arange(K)
            rp = (37 * i + r - 37 * j) % W
            shared = rp < K
            hit = row[shared] == rp[shared]
            assert hit.mean() > 0.97, (i, j, hit.mean())
            assert (row[~shared] < 0).mean() > 0.97
        # and the oracle outright on three pairs (first, one in the middle, last pair of the list)
        for (i, j) in ((0, 1), (731, 1408), (n - 2, n - 1)):
            p = i * (2 * n - i - 1) // 2 + (j - i - 1)
            exp, ec = orc.match_grid([block[i].cpu().numpy(), block[j].cpu().numpy()], np.array([[0, 1]], np.int32), threads=8)
            assert np.array_equal(out[p].cpu().numpy(), exp[0]) and int(cnt[p].item()) == int(ec[0])
    finally:
        m.clear()

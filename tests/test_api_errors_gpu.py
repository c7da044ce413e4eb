"""GPU suite: error behaviour of the C ABI (status codes, never a crash, message available)."""
import ctypes as C

import numpy as np
import pytest

from reconstructor_amd import _lib, synth_ba

pytestmark = pytest.mark.gpu


def test_matcher_errors(gpu_ctx):
    from reconstructor_amd.matcher import HipL2Matcher
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    a = np.random.default_rng(0).standard_normal((70, 32)).astype(np.float32)
    m.upload(0, a)
    m.upload(1, a[:50])
    with pytest.raises(_lib.RcnError) as e:                       # image 5 is not resident
        m.match_grid(np.array([[0, 5]], np.int32), 70)
    assert e.value.code == -5 and "not resident" in str(e.value)
    with pytest.raises(_lib.RcnError) as e:                       # out_stride smaller than K of the query image
        m.match_grid(np.array([[0, 1]], np.int32), 10)
    assert e.value.code == -1
    with pytest.raises(_lib.RcnError) as e:                       # D differs from the resident images
        m.upload(2, np.zeros((5, 16), np.float32))
    assert e.value.code == -1
    out, counts = m.match_grid(np.zeros((0, 2), np.int32), 70)    # empty grid is fine
    assert out.shape[0] == 0
    assert gpu_ctx.lib.rcn_desc_count(gpu_ctx.h) == 2
    m.clear()
    assert gpu_ctx.lib.rcn_desc_count(gpu_ctx.h) == 0
    rc = gpu_ctx.lib.rcn_match_pair(gpu_ctx.h, None, 4, None, 4, 8, 0.7, None, None)   # null pointers
    assert rc == -1
    import torch
    buf = torch.zeros(2 * 8 * 32 + 4, device="cuda")
    with pytest.raises(_lib.RcnError) as e:                       # borrowed block not 16-byte aligned
        m.upload_batch_device(0, 2, buf.data_ptr() + 4, 8, 32)
    assert e.value.code == -1 and "aligned" in str(e.value)
    m.upload_batch_device(0, 2, buf.data_ptr(), 8, 32)
    m.clear()


def test_ba_errors(gpu_ctx):
    from reconstructor_amd import ba
    sc = synth_ba.make_scene(3, 10, obs_per_point=3, seed=1)
    bad = dict(sc)
    bad["obs_pt"] = sc["obs_pt"][::-1].copy()                     # not landmark-major
    with pytest.raises(_lib.RcnError) as e:
        ba.solve_scene(gpu_ctx, bad)
    assert e.value.code == -1 and "landmark-major" in str(e.value)
    bad = dict(sc)
    bad["obs_cam"] = sc["obs_cam"].copy(); bad["obs_cam"][0] = 9  # camera index out of range
    with pytest.raises(_lib.RcnError):
        ba.solve_scene(gpu_ctx, bad)
    rc = gpu_ctx.lib.rcn_ba_solve(gpu_ctx.h, None, None, None)
    assert rc == -1
    # a degenerate problem (no observations) terminates cleanly
    P, I, X, s = ba.solve_flat(gpu_ctx, sc["poses"], sc["intrinsics"], sc["points"],
                               np.zeros((0, 2)), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert s["iterations"] == 0 and np.array_equal(P, sc["poses"])


def test_plugin_calls_from_four_threads(gpu_ctx):
    """The reference calls matchFeatures / estimateFundamental from 4 OpenMP threads on one shared
    plugin object (SequentialReconstructor.cpp:202): the per-pair entry points must be re-entrant."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orc, orc_fmat
    from reconstructor_amd import fmat, synth, synth_fmat
    from reconstructor_amd.matcher import HipL2Matcher
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    ims = synth.descriptor_set("sift", 6, [180, 220, 150, 260, 200, 170], n_world=500, seed=31)
    jobs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    two = [synth_fmat.two_view(120 + 10 * k, 0.3, seed=k) for k in range(len(jobs))]

    def work(k):
        i, j = jobs[k]
        got = m.match_pair(ims[i], ims[j])
        mask, cnt = fmat.estimate_fundamental_inliers(gpu_ctx, two[k][0], two[k][1])
        return got, mask, cnt
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(work, range(len(jobs))))
    for k, (i, j) in enumerate(jobs):
        exp, cnt = orc.match_pair(ims[i], ims[j])
        assert np.array_equal(res[k][0], exp) and (res[k][0] >= 0).sum() == cnt
        m0, c0, _ = orc_fmat.filter_pair(two[k][0], two[k][1])
        assert np.array_equal(res[k][1], m0) and res[k][2] == c0


def test_shard_and_list_errors(gpu_ctx):
    """Misuse of the sharded-grid / host-list / session entry points returns a status, never a crash."""
    import torch
    from reconstructor_amd import ba, pairgrid
    lib = gpu_ctx.lib
    sh = pairgrid.Shard(gpu_ctx, 0, 1, pairgrid.unique_id())
    try:
        with pytest.raises(_lib.RcnError) as e:                   # exchange before reserve
            sh.exchange(None)
        assert e.value.code == -1 and "reserve" in str(e.value)
        with pytest.raises(_lib.RcnError):                        # match before exchange
            sh.match(0.7)
        gpu_ctx.check(lib.rcn_desc_clear(gpu_ctx.h))
        sh.reserve(3, 64, 32)
        with pytest.raises(_lib.RcnError) as e:                   # rows do not fit the slot
            sh.put_image(0, np.zeros((65, 32), np.float32))
        assert e.value.code == -1
        with pytest.raises(_lib.RcnError):                        # not an image of the grid
            sh.put_image(3, np.zeros((4, 32), np.float32))
        with pytest.raises(_lib.RcnError):                        # a row count beyond the slot
            sh.exchange(torch.zeros((3, 64, 32), device="cuda").data_ptr(), [64, 65, 1])
        with pytest.raises(_lib.RcnError) as e:                   # lists without a match into the ctx's own tables
            gpu_ctx.check(lib.rcn_shard_lists(sh.h, np.zeros(4, np.int64).ctypes.data, None, 0, C.byref(C.c_int64())))
        assert e.value.code == -1
        assert lib.rcn_shard_reserve(sh.h, 0, 64, 32, None) == -1 and lib.rcn_shard_reserve(sh.h, 3, 0, 32, None) == -1
        assert lib.rcn_shard_create(gpu_ctx.h, 1, 1, (C.c_uint8 * 128)(), C.byref(C.c_void_p())) == -1     # rank >= world
    finally:
        sh.close()
    total = C.c_int64()
    assert lib.rcn_match_compact_begin(gpu_ctx.h, None, 8, None, 3, np.zeros(4, np.int64).ctypes.data, None, 0, C.byref(total)) == -1
    assert lib.rcn_match_compact_begin(gpu_ctx.h, None, 8, None, 0, np.zeros(1, np.int64).ctypes.data, None, 0, C.byref(total)) == 0 and total.value == 0
    assert lib.rcn_match_compact_wait(gpu_ctx.h) == 0
    ses = ba.BaSession(gpu_ctx)
    try:
        with pytest.raises(_lib.RcnError) as e:                   # nothing to solve
            ses.solve()
        assert e.value.code == -1 and "camera" in str(e.value)
        ses.add_camera(np.zeros(6), np.ones(6))
        with pytest.raises(_lib.RcnError):                        # landmark index out of range
            ses.add_observations([0], [0], [[1, 2]])
        ses.add_points(np.zeros((2, 3)))
        with pytest.raises(_lib.RcnError):                        # camera index out of range
            ses.add_observations([0], [1], [[1, 2]])
        with pytest.raises(_lib.RcnError) as e:                   # no sweep yet
            ses.remove_outliers()
        assert "sweep" in str(e.value)
        assert ses.counts() == (1, 2, 0)
    finally:
        ses.close()
    assert lib.rcn_desc_sample_device(gpu_ctx.h, None, 1, 1, 1, 4, 4, None, 5, 300, None) == -1            # D > 256

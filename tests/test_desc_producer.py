"""The descriptor producer contract: oracle/desc_oracle.c (processDescriptors of the reference, FeatureSuperPoint.cpp:183-211)
against an independent numpy transcription (CPU tier); the HIP kernel against the oracle, and rows produced on the
device straight into the matcher's landing buffer (GPU tier)."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc


def _map_and_keypoints(seed, Hc=21, Wc=30, C_=256, K=500):
    rng = np.random.default_rng(seed)
    m = rng.standard_normal((C_, Hc, Wc)).astype(np.float32)
    m[:, 3, 4] *= np.float32(1e-6)              # a cell of tiny values next to ordinary ones: summation order matters
    m[5, 7, 9] = np.float32(300.0)              # one dominant channel
    kp = np.stack([rng.integers(0, 8 * Wc, K), rng.integers(0, 8 * Hc, K)], 1).astype(np.int32)
    kp[0] = (8 * 4 + 3, 8 * 3 + 7)
    kp[1] = (8 * 9, 8 * 7)
    return m, kp


def test_oracle_against_a_numpy_transcription():
    m, kp = _map_and_keypoints(1)
    got = orc.desc_sample(m, kp)
    for k, (x, y) in enumerate(kp):
        v = m[:, y // 8, x // 8][:256]
        sq = (v * v).astype(np.float32)                     # fp32 products (`descElem * descElem`)
        s = np.float64(0.0)
        for p in sq:
            s = s + np.float64(p)                           # double sum in ascending order
        want = (v.astype(np.float64) / np.sqrt(s)).astype(np.float32)
        assert got[k].tobytes() == want.tobytes(), k
    assert np.allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # the channel-last copy of the map gives the same rows
    assert np.array_equal(orc.desc_sample(np.ascontiguousarray(m.transpose(1, 2, 0)), kp, channel_last=True), got)


@pytest.mark.gpu
@pytest.mark.parametrize("channel_last", [False, True])
def test_device_rows_equal_the_oracle(gpu_ctx, channel_last):
    import torch
    m, kp = _map_and_keypoints(2, K=1300)
    Cn, Hc, Wc = m.shape
    src = np.ascontiguousarray(m.transpose(1, 2, 0)) if channel_last else m
    strides = (1, Wc * Cn, Cn) if channel_last else (Hc * Wc, Wc, 1)
    dm, dk = torch.from_numpy(src).cuda(), torch.from_numpy(kp).cuda()
    for D in (256, 100):
        out = torch.zeros((len(kp), D), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        gpu_ctx.check(gpu_ctx.lib.rcn_desc_sample_device(gpu_ctx.h, dm.data_ptr(), *strides, Hc, Wc, dk.data_ptr(), len(kp), D, out.data_ptr()))
        gpu_ctx.check(gpu_ctx.lib.rcn_synchronize(gpu_ctx.h))
        want = orc.desc_sample(src, kp, D=D, channel_last=channel_last)
        assert out.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.gpu
def test_detector_writes_straight_into_the_landing_buffer(gpu_ctx):
    """Producer contract end to end: per image, rows sampled on the device land in that image's slot of the shard's
    landing buffer (no host copy, no featDescToCV), then exchange + match; equal to the oracle on the oracle's rows."""
    import torch
    from reconstructor_amd import pairgrid
    from reconstructor_amd.matcher import all_pairs
    n, K, D = 4, 700, 256
    maps, kps = zip(*[_map_and_keypoints(10 + i, K=K) for i in range(n)])
    # images share keypoint cells with their neighbour so that matches exist
    maps = [maps[0]] + [(0.9 * maps[0] + 0.1 * mm).astype(np.float32) for mm in maps[1:]]
    rows = [orc.desc_sample(mm, kk) for mm, kk in zip(maps, kps)]
    exp, ec = orc.match_grid(rows, all_pairs(n), threads=4)
    sh = pairgrid.Shard(gpu_ctx, 0, 1, pairgrid.unique_id())
    try:
        gpu_ctx.check(gpu_ctx.lib.rcn_desc_clear(gpu_ctx.h))
        slot = sh.reserve(n, K, D)
        Cn, Hc, Wc = maps[0].shape
        keep = []
        for i in range(n):
            dm, dk = torch.from_numpy(maps[i]).cuda(), torch.from_numpy(kps[i]).cuda()
            keep += [dm, dk]
            torch.cuda.synchronize()
            gpu_ctx.check(gpu_ctx.lib.rcn_desc_sample_device(gpu_ctx.h, dm.data_ptr(), Hc * Wc, Wc, 1, Hc, Wc, dk.data_ptr(), K, D,
                                                             C.c_void_p(slot + i * K * D * 4)))
        sh.exchange(None, None)
        P = n * (n - 1) // 2
        out = torch.empty((P, K), dtype=torch.int32, device="cuda")
        cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        sh.match(0.7, out.data_ptr(), K, cnt.data_ptr())
        gpu_ctx.check(gpu_ctx.lib.rcn_synchronize(gpu_ctx.h))
        assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(cnt.cpu().numpy(), ec) and ec.sum() > 0
    finally:
        sh.close()


@pytest.mark.gpu
def test_keypoints_outside_the_map_read_nothing_and_are_reported(gpu_ctx):
    """The reference indexes the descriptor tensor with keypoint / 8 (FeatureSuperPoint.cpp:191-195), which throws outside
    the map; here such a keypoint reads nothing, its row is zeros, and rcn_desc_sample_errors reports it."""
    import ctypes as C
    import torch
    m, kp = _map_and_keypoints(3, K=64)
    Cn, Hc, Wc = m.shape
    kp = kp.copy()
    kp[5] = (8 * Wc, 3)            # one cell right of the map
    kp[9] = (4, 8 * Hc + 7)        # below it
    kp[11] = (-1, 2)               # negative
    dm, dk = torch.from_numpy(m).cuda(), torch.from_numpy(kp).cuda()
    out = torch.full((len(kp), 256), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n_bad = C.c_int32(-1)
    assert gpu_ctx.lib.rcn_desc_sample_errors(gpu_ctx.h, C.byref(n_bad)) == 0 and n_bad.value == 0
    gpu_ctx.check(gpu_ctx.lib.rcn_desc_sample_device(gpu_ctx.h, dm.data_ptr(), Hc * Wc, Wc, 1, Hc, Wc, dk.data_ptr(), len(kp), 256, out.data_ptr()))
    assert gpu_ctx.lib.rcn_desc_sample_errors(gpu_ctx.h, C.byref(n_bad)) == -1 and n_bad.value == 3
    assert b"outside the descriptor map" in gpu_ctx.lib.rcn_last_error(gpu_ctx.h)
    got = out.cpu().numpy()
    ok = np.ones(len(kp), bool)
    ok[[5, 9, 11]] = False
    assert (got[~ok] == 0).all()
    assert got[ok].tobytes() == orc.desc_sample(m, kp[ok]).tobytes()
    assert gpu_ctx.lib.rcn_desc_sample_errors(gpu_ctx.h, C.byref(n_bad)) == 0 and n_bad.value == 0      # read clears

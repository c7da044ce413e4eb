"""The store file (rcn_store_*: features + matches on disk, the cache the reference lists as a TODO, README.md:39):
round trips are bit-exact, damage is detected (CPU tier); a matching stage resumed from the file reproduces the
tables (GPU tier)."""
import os

import numpy as np
import pytest

from reconstructor_amd import _lib, store, synth


def _contents(seed=3, with_coords=True):
    rng = np.random.default_rng(seed)
    Ks = [40, 0, 17, 64]
    ims = synth.descriptor_set("sift", 4, [max(k, 1) for k in Ks], n_world=200, seed=seed)
    ims = [im[:k] for im, k in zip(ims, Ks)]
    coords = [rng.integers(0, 512, (k, 2)).astype(np.int32) for k in Ks] if with_coords else None
    pairs = np.array([(0, 2), (0, 3), (2, 3), (3, 0)], np.int32)
    counts = [5, 0, 9, 3]
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    qt = np.stack([np.concatenate([np.sort(rng.choice(Ks[a], c, replace=False)) for (a, b), c in zip(pairs, counts)]),
                   np.concatenate([rng.choice(Ks[b], c, replace=False) for (a, b), c in zip(pairs, counts)])], 1).astype(np.int32)
    return ims, coords, pairs, offsets, qt


@pytest.mark.parametrize("with_coords", [True, False])
def test_round_trip_is_bit_exact(tmp_path, with_coords):
    ims, coords, pairs, offsets, qt = _contents(with_coords=with_coords)
    path = tmp_path / "scene.rcn"
    store.save(path, ims, coords, pairs, offsets, qt, img_ids=[10, 11, 12, 13])
    assert not os.path.exists(str(path) + ".tmp")
    with store.Store(path) as s:
        assert (s.n_images, s.D, s.has_coords, s.n_pairs) == (4, 128, with_coords, 4)
        assert list(s.img_ids) == [10, 11, 12, 13] and list(s.img_K) == [40, 0, 17, 64]
        for a, b in zip(s.images, ims):
            assert a.shape == b.shape and a.tobytes() == b.tobytes()
        if with_coords:
            assert all(np.array_equal(a, b) for a, b in zip(s.coords, coords))
        assert np.array_equal(s.pairs, pairs) and np.array_equal(s.offsets, offsets) and np.array_equal(s.qt, qt)
    # saving what was loaded gives the same bytes: the format has no hidden state
    store.save(tmp_path / "again.rcn", ims, coords, pairs, offsets, qt, img_ids=[10, 11, 12, 13])
    assert open(path, "rb").read() == open(tmp_path / "again.rcn", "rb").read()


def test_features_only_and_empty(tmp_path):
    ims, coords, *_ = _contents()
    store.save(tmp_path / "f.rcn", ims, coords)
    with store.Store(tmp_path / "f.rcn") as s:
        assert s.n_pairs == 0 and len(s.qt) == 0 and len(s.images) == 4
    store.save(tmp_path / "e.rcn", [])
    with store.Store(tmp_path / "e.rcn") as s:
        assert s.n_images == 0 and s.n_pairs == 0


def test_damage_is_detected(tmp_path):
    ims, coords, pairs, offsets, qt = _contents()
    path = tmp_path / "scene.rcn"
    store.save(path, ims, coords, pairs, offsets, qt)
    raw = bytearray(open(path, "rb").read())
    for pos, what in ((0, "magic"), (8, "version"), (len(raw) // 2, "payload"), (len(raw) - 1, "checksum")):
        bad = bytearray(raw)
        bad[pos] ^= 0x40
        open(tmp_path / "bad.rcn", "wb").write(bad)
        with pytest.raises(_lib.RcnError) as e:
            store.Store(tmp_path / "bad.rcn")
        assert e.value.code == -8, what                       # RCN_ERR_IO
    open(tmp_path / "short.rcn", "wb").write(raw[:len(raw) - 9])
    with pytest.raises(_lib.RcnError):
        store.Store(tmp_path / "short.rcn")
    with pytest.raises(_lib.RcnError):
        store.Store(tmp_path / "missing.rcn")
    # inconsistent contents are refused at save time
    with pytest.raises(_lib.RcnError):
        store.save(tmp_path / "x.rcn", ims, coords, pairs, offsets[::-1].copy(), qt)


@pytest.mark.gpu
def test_matching_stage_resumes_from_the_file(gpu_ctx, tmp_path):
    """features -> match -> lists -> file; a fresh ctx state loads the file, matches again: same tables, and the
    stored lists are what the second run produces."""
    import ctypes as C
    from oracle import orc
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    ims = synth.descriptor_set("sift", 5, [300, 120, 0, 513, 64], n_world=700, seed=21)
    rng = np.random.default_rng(1)
    coords = [rng.integers(0, 512, (len(im), 2)).astype(np.int32) for im in ims]
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    for i, im in enumerate(ims):
        if len(im):
            m.upload(i, im)
        else:
            gpu_ctx.check(gpu_ctx.lib.rcn_desc_upload(gpu_ctx.h, i, None, 0, 128))
    pairs = all_pairs(5)
    out, cnt = m.match_grid(pairs, 513)
    exp, ec = orc.match_grid([im if len(im) else np.zeros((0, 128), np.float32) for im in ims], pairs, threads=2)
    assert np.array_equal(out, exp)
    offsets = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    qt = np.array([(q, out[p][q]) for p in range(len(pairs)) for q in np.nonzero(out[p] >= 0)[0]], np.int32).reshape(-1, 2)
    store.save(tmp_path / "stage.rcn", ims, coords, pairs, offsets, qt)
    m.clear()
    with store.Store(tmp_path / "stage.rcn") as s:
        s.upload(gpu_ctx)
        assert gpu_ctx.lib.rcn_desc_count(gpu_ctx.h) == 5
        out2, cnt2 = m.match_grid(s.pairs, 513)
        assert np.array_equal(out2, out) and np.array_equal(cnt2, cnt)
        assert np.array_equal(np.diff(s.offsets), cnt2)

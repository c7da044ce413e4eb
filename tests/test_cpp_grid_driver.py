"""The C++ multi-GPU pair-grid driver (reconstructor_amd/host/HipPairGridDriver.h: one host thread + one
rcn_shard per GPU, RCCL behind include/rcn.h) driven like SequentialReconstructor::matchFeatures and checked
against a restatement of that loop over the CPU oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import orc, orc_fmat
from reconstructor_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "grid_driver_test")


def test_driver_compiles_without_gpu():
    """CPU tier: the driver builds against include/rcn.h with plain g++ -- no HIP, no RCCL headers."""
    import __graft_entry__ as g
    g.build_cpp_tests()
    assert os.path.exists(BIN)


def reference_loop(ims, coords=None):
    """SequentialReconstructor.cpp:202-279 in sequential order: (i, j) is matched with query = i unless the inverse
    pair already has an entry, in which case that entry is inverted; a pair that stored nothing leaves no entry, so its
    inverse is matched in its own right.  coords given = matchFeatures(filter = true): a pair with at least 7 matches
    keeps the inliers of the fundamental-matrix search only, and stores NOTHING when no model is found (:237-269)."""
    n = len(ims)
    fm = {}
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            if (j, i) in fm:
                fm[(i, j)] = {t: q for q, t in fm[(j, i)].items()}
                continue
            out, cnt = orc.match_pair(ims[i], ims[j]) if len(ims[i]) and len(ims[j]) else (np.zeros(0, np.int32), 0)
            q = np.nonzero(out >= 0)[0]
            if coords is not None and cnt >= 7:
                mask, count, _ = orc_fmat.filter_grid(np.array([0, cnt], np.int32), coords[i][q], coords[j][out[q]])
                if count[0] < 0:
                    continue                               # inlierMatchIds.size() == 0 -> continue (:252-255)
                q = q[mask]
            if len(q):
                fm[(i, j)] = {int(a): int(out[a]) for a in q}
    return fm


def _run_driver(tmp_path, ims, coords, D, *extra):
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("ii", len(ims), D))
        for im, xy in zip(ims, coords):
            f.write(struct.pack("i", len(im)))
            f.write(np.ascontiguousarray(im, np.float32).tobytes())
            f.write(np.ascontiguousarray(xy, np.int32).tobytes())
    r = subprocess.run([BIN, str(inp), str(outp), *[str(e) for e in extra]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.frombuffer(open(outp, "rb").read(), np.int32)
    world, n_entries = raw[0], raw[1]
    assert world >= 1
    got, pos = {}, 2
    for _ in range(n_entries):
        i, j, c = raw[pos:pos + 3]
        qt = raw[pos + 3:pos + 3 + 2 * c].reshape(-1, 2)
        got[(int(i), int(j))] = {int(q): int(t) for q, t in qt}
        pos += 3 + 2 * c
    return got, r


@pytest.mark.gpu
def test_cpp_grid_driver_equals_the_reference_loop(tmp_path):
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    # ragged K; image 3 shares no world point with the others' pool -> mostly empty pairs both ways; image 5 is empty
    ims = synth.descriptor_set("sift", 7, [300, 420, 64, 200, 513, 0, 97], n_world=900, seed=13)
    ims[3] = synth.descriptor_set("sift", 1, 200, n_world=400, seed=999)[0]
    ims[6] = ims[0][7:8].copy()    # ONE keypoint: as a train image it yields nothing (K2 < 2), as a query it matches
    coords = [np.zeros((len(im), 2), np.int32) for im in ims]
    got, _ = _run_driver(tmp_path, ims, coords, 128)
    exp = reference_loop(ims)
    assert set(got) == set(exp)
    assert all(got[k] == exp[k] for k in exp)
    # the second pass really ran: some pair has an entry one way only
    assert (6, 0) in exp and (0, 6) not in exp


@pytest.mark.gpu
def test_cpp_grid_driver_whole_pair_loop_with_the_filter(tmp_path):
    """matchFeatures(filter = true) from C++ -- match, epipolar filter on the table in HBM, lists, second pass --
    against a sequential restatement of SequentialReconstructor.cpp:199-279 over the two CPU oracles, bit-equal."""
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    # image 4 is small enough for every pair with it to stay below 7 matches (unfiltered); no pair has 8..13 matches,
    # where the LMedS branch picks among exact fits by rounding noise and only the shape of the answer is comparable
    # between two implementations
    ims, coords, _ = synth.scene_set("sift", 7, [300, 320, 280, 310, 12, 300, 0], n_world=900, seed=5)
    coords[5][:] = 91          # every sample of a pair with image 5 is degenerate: no model -> nothing stored -> reverse pass
    store_path = tmp_path / "loop.rcnstore"
    got, r = _run_driver(tmp_path, ims, coords, 128, 0, 1, -1, store_path)      # ... and shard + filter + STORE in the same C++ run
    assert "store round trip ok" in r.stdout and store_path.exists()
    from reconstructor_amd import store as rstore
    with rstore.Store(store_path) as back:                                      # the Python side reads the same file
        assert back.n_images == len(ims) and back.has_coords and back.n_pairs == len(got)
        assert all(np.array_equal(back.images[i], ims[i]) and np.array_equal(back.coords[i], coords[i]) for i in range(len(ims)))
        stored = {(int(a), int(b)): {int(q): int(t) for q, t in back.qt[int(back.offsets[p]):int(back.offsets[p + 1])]} for p, (a, b) in enumerate(back.pairs)}
        assert stored == got
    exp = reference_loop(ims, coords)
    plain = reference_loop(ims)
    assert not any(8 <= len(v) <= 13 for v in plain.values())
    assert set(got) == set(exp)
    assert [k for k in exp if got[k] != exp[k]] == []
    assert sum(len(v) for v in exp.values()) < sum(len(v) for v in plain.values())       # the filter removed something
    assert any(k not in exp for k in plain)                                              # and dropped whole pairs (no model)
    assert any(0 < len(v) < 7 for v in exp.values())                                     # pairs below 7 matches pass unfiltered


@pytest.mark.gpu
def test_cpp_grid_driver_local_failure_fails_everywhere_and_recovers(tmp_path):
    """Fault injection at world size 1 (the only size a one-GPU box has): a rank that reports a local failure still
    enters the exchange, whose status vote fails the call; the driver throws instead of hanging and works afterwards."""
    ims = synth.descriptor_set("sift", 4, [120, 90, 100, 64], n_world=300, seed=3)
    coords = [np.zeros((len(im), 2), np.int32) for im in ims]
    got, r = _run_driver(tmp_path, ims, coords, 128, 0, 0, 0)
    assert "injected:" in r.stderr and "rcn_shard_exchange" in r.stderr
    exp = reference_loop(ims)
    assert got == exp

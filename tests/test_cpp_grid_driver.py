"""The C++ multi-GPU pair-grid driver (reconstructor_amd/host/HipPairGridDriver.h: one host thread + one
rcn_shard per GPU, RCCL behind include/rcn.h) driven like SequentialReconstructor::matchFeatures and checked
against a restatement of that loop over the CPU oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import orc
from reconstructor_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "grid_driver_test")


def test_driver_compiles_without_gpu():
    """CPU tier: the driver builds against include/rcn.h with plain g++ -- no HIP, no RCCL headers."""
    import __graft_entry__ as g
    g.build_cpp_tests()
    assert os.path.exists(BIN)


def reference_loop(ims):
    """SequentialReconstructor.cpp:202-279 in sequential order with the filter off: (i, j) is matched with
    query = i unless the inverse pair already has an entry, in which case that entry is inverted; a pair that
    stored nothing leaves no entry, so its inverse is matched in its own right."""
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
            if cnt:
                fm[(i, j)] = {int(q): int(out[q]) for q in np.nonzero(out >= 0)[0]}
    return fm


@pytest.mark.gpu
def test_cpp_grid_driver_equals_the_reference_loop(tmp_path):
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    # ragged K; image 3 shares no world point with the others' pool -> mostly empty pairs both ways; image 5 is empty
    ims = synth.descriptor_set("sift", 7, [300, 420, 64, 200, 513, 0, 97], n_world=900, seed=13)
    ims[3] = synth.descriptor_set("sift", 1, 200, n_world=400, seed=999)[0]
    ims[6] = ims[0][7:8].copy()    # ONE keypoint: as a train image it yields nothing (K2 < 2), as a query it matches
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("ii", len(ims), 128))
        for im in ims:
            f.write(struct.pack("i", len(im)))
            f.write(np.ascontiguousarray(im, np.float32).tobytes())
    r = subprocess.run([BIN, str(inp), str(outp)], capture_output=True, text=True, timeout=300)
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
    exp = reference_loop(ims)
    assert set(got) == set(exp)
    assert all(got[k] == exp[k] for k in exp)
    # the second pass really ran: some pair has an entry one way only
    assert (6, 0) in exp and (0, 6) not in exp

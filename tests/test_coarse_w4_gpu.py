"""GPU suite: the second form of K1 (k_coarse_w4, csrc/coarse_w4.h: one wave per SIMD, query fragments in AGPRs, hand-scheduled
stream).  It is not the shipping kernel -- on MI355X both forms land on the same power-limited throughput
-- and exists only in the diagnostic build; it is kept correct: a child process runs it (RCN_COARSE_W4=1) against the oracle,
repeatedly, because the bug class of such a stream is a race that shows on some launches only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from oracle import orc
from reconstructor_amd import _lib, synth
from reconstructor_amd.matcher import HipL2Matcher, all_pairs
assert b"DIAGNOSTIC" in _lib.load().rcn_version()
m = HipL2Matcher(device=0)
for kind, n, K in (("superpoint", 10, 2048), ("sift", 8, [1500, 700, 2048, 64, 513, 1, 900, 1300]), ("superpoint", 5, 300),
                   ("orb", 6, 900), ("sift64", 6, 700)):
    ims = synth.descriptor_set("sift" if kind == "sift64" else kind, n, K, n_world=4096, seed=11)
    if kind == "sift64":                      # a 64-d descriptor (DP = 64 instantiation): the first half of the SIFT rows
        ims = [np.ascontiguousarray(im[:, :64]) for im in ims]
    pairs = all_pairs(n)
    exp, ec = orc.match_grid(ims, pairs, threads=8)
    m.clear()
    for i, im in enumerate(ims):
        m.upload(i, im)
    for rep in range(6):
        out, cnt = m.match_grid(pairs, exp.shape[1])
        assert np.array_equal(out, exp) and np.array_equal(cnt, ec), (kind, rep)
    assert m.stats()["used_mfma_path"] == 1
print("OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("switch,value", [("RCN_COARSE_W4", "1"), ("RCN_COARSE_S16", "0"), ("RCN_COARSE_S16", "1")])
def test_other_forms_of_k1_equal_the_oracle_on_every_launch(switch, value):
    """RCN_COARSE_W4: k_coarse_w4; RCN_COARSE_S16 = 0 / 1: k_coarse_top2 forced onto v_mfma_f32_32x32x16_f16 /
    v_mfma_f32_16x16x32_f16 at every D (the shipping library picks per D)."""
    diag = os.path.join(ROOT, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "run __graft_entry__.build() first"
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=dict(os.environ, RCN_LIB=diag, **{switch: value}),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr

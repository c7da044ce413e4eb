"""GPU suite: the exact stages behind the coarse pass -- the middle tier (fp32 sweep shared by the rows of a train image ->
candidates -> fp64 chain), K2b behind it, and the pipeline chunks that bound the workspace.  The shipping library takes these
paths for a fraction of a percent of the rows and for grids of hundreds of thousands of pairs; the diagnostic build can force
them on small grids (RCN_FORCE_EXACT: no coarse pass, every row enters the tier without a threshold -- two sweeps; RCN_MID_ROWS:
the tier's row budget, so that the rest reaches K2b; RCN_CHUNK_ROWS: rows per pipeline chunk), in a child process, against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc
from reconstructor_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
from oracle import orc
from reconstructor_amd import _lib, synth
from reconstructor_amd.matcher import HipL2Matcher, all_pairs
assert b"DIAGNOSTIC" in _lib.load().rcn_version()
m = HipL2Matcher(device=0)
res = []
rng = np.random.default_rng(3)
for kind, n, K in (("superpoint", 7, [700, 300, 1024, 64, 513, 2, 900]), ("sift", 6, [1500, 700, 33, 640, 1, 257]), ("orb", 6, 400)):
    ims = synth.descriptor_set(kind, n, K, n_world=1500, seed=23)
    for im in ims:                               # exact ties and near-ties for the top-2
        if len(im) >= 8:
            a, b, c, d = rng.integers(0, len(im), 4)
            im[a] = im[b]
            im[c] = im[d] + (rng.standard_normal(im.shape[1]) * 1e-4).astype(np.float32)
    ims[1] = ims[1] * np.float32(2.0 ** 30)      # one image out of scale: BIG rows as queries and as train rows
    pairs = np.concatenate([all_pairs(n), all_pairs(n)[:, ::-1]]).astype(np.int32)
    exp, ec = orc.match_grid(ims, pairs, threads=8)
    m.clear()
    for i, im in enumerate(ims):
        m.upload(i, im)
    for rep in range(2):
        out, cnt = m.match_grid(pairs, exp.shape[1])
        assert np.array_equal(out, exp) and np.array_equal(cnt, ec), (kind, rep, np.argwhere(out != exp)[:5].tolist())
    st = m.stats()
    res.append({k: int(st[k]) for k in ("rows_total", "rows_exact_fallback", "rows_brute_force", "chunks", "used_mfma_path")})
print(json.dumps(res))
"""


@pytest.mark.parametrize("env", [{}, {"RCN_FORCE_EXACT": "1"}, {"RCN_FORCE_EXACT": "1", "RCN_MID_ROWS": "1000"},
                                 {"RCN_CHUNK_ROWS": "4096"}, {"RCN_CHUNK_ROWS": "3000", "RCN_MID_ROWS": "50"},
                                 {"RCN_FORCE_EXACT": "1", "RCN_CHUNK_ROWS": "5000"}],
                         ids=["shipping-settings", "every-row-through-the-tier", "tier-budget-exceeded", "many-chunks", "chunks-and-a-tiny-tier", "no-coarse-pass-in-chunks"])
def test_exact_tiers_and_chunks_equal_the_oracle(env):
    import json
    diag = os.path.join(ROOT, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "run __graft_entry__.build() first"
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=dict(os.environ, RCN_LIB=diag, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for d in res:
        if "RCN_FORCE_EXACT" in env:
            assert d["used_mfma_path"] == 0 and d["rows_exact_fallback"] > 0
            if "RCN_MID_ROWS" in env:
                assert d["rows_brute_force"] > 0                  # beyond the tier's budget: K2b
            elif "RCN_CHUNK_ROWS" not in env:
                # the tier resolves most rows itself; what it passes on here are the queries against the image scaled by 2^30 when its
                # rows have (nearly) equal norms: all their fp32 distances agree to 1e-9, far inside the fp32 error band
                assert d["rows_brute_force"] < d["rows_exact_fallback"] // 3
        if "RCN_CHUNK_ROWS" in env:
            assert d["chunks"] > 3
        else:
            assert d["chunks"] == 1


def test_chunks_and_the_deferred_pass_do_not_change_what_reaches_the_tier():
    """Pipeline chunks hand their rows to ONE deferred pass of the middle tier (k_mid_defer); a deferred list that is too small
    sends a chunk's rows through the tier at once.  Either way the same rows leave the re-rank and the same rows reach K2b as in
    one chunk (the tables are compared with the oracle inside the script)."""
    import json
    diag = os.path.join(ROOT, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "run __graft_entry__.build() first"
    got = {}
    for name, env in (("one", {}), ("chunks", {"RCN_CHUNK_ROWS": "4096"}), ("tiny-list", {"RCN_CHUNK_ROWS": "4096", "RCN_MID_ROWS": "40"})):
        r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=dict(os.environ, RCN_LIB=diag, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        got[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for a, b, c in zip(got["one"], got["chunks"], got["tiny-list"]):
        assert a["chunks"] == 1 and b["chunks"] > 3 and c["chunks"] > 3
        assert a["rows_total"] == b["rows_total"] == c["rows_total"]
        assert a["rows_exact_fallback"] == b["rows_exact_fallback"] == c["rows_exact_fallback"]
        assert a["rows_brute_force"] == b["rows_brute_force"]          # (a list of 40 rows leaves more to K2b: only the tables must agree there)


def test_mid_tier_on_a_generic_descriptor_length(gpu_ctx):
    """D = 320 (> 256: no MFMA path, D % 4 == 0): every row goes through the middle tier of the SHIPPING library."""
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    rng = np.random.default_rng(1)
    ims = [rng.standard_normal((k, 320)).astype(np.float32) for k in (300, 150, 2, 513)]
    ims[3][7] = ims[3][9]
    pairs = all_pairs(4)
    exp, ec = orc.match_grid(ims, pairs, threads=4)
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    for i, im in enumerate(ims):
        m.upload(i, im)
    out, cnt = m.match_grid(pairs, 513)
    st = m.stats()
    m.clear()
    assert np.array_equal(out, exp) and np.array_equal(cnt, ec)
    assert st["used_mfma_path"] == 0 and st["rows_exact_fallback"] == st["rows_total"] and st["rows_brute_force"] < st["rows_total"] // 10

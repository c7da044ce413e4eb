"""tools/w4_stress.py -- repeat one grid call and compare every run with the oracle (race hunting aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import orc
from reconstructor_amd import synth
from reconstructor_amd.matcher import HipL2Matcher, all_pairs

n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ims = synth.descriptor_set("superpoint", n, K, n_world=4 * K, seed=1234)
pairs = all_pairs(n)
exp, ec = orc.match_grid(ims, pairs, threads=16)
m = HipL2Matcher(device=0)
for i, im in enumerate(ims):
    m.upload(i, im)
bad = 0
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 20):
    out, cnt = m.match_grid(pairs, K)
    ok = np.array_equal(out, exp)
    if not ok:
        bad += 1
        d = np.argwhere(out != exp)
        print("run", it, "differs in", len(d), "entries; pairs", sorted(set(d[:, 0]))[:10], "rows", d[:5, 1], "stats", m.stats()["rows_exact_fallback"])
print("bad runs:", bad)

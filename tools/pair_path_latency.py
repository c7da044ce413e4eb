import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from reconstructor_amd import synth
from reconstructor_amd.matcher import HipL2Matcher
from oracle import orc
m = HipL2Matcher()
ims = synth.descriptor_set("sift", 12, [1500 + 37 * i for i in range(12)], n_world=4000, seed=3)
m.match_pair(ims[0], ims[1])
t0 = time.perf_counter(); n = 0
for i in range(12):
    for j in range(i + 1, 12):
        out = m.match_pair(ims[i], ims[j]); n += 1
dt = time.perf_counter() - t0
print("per-pair plugin path: %d calls, %.2f ms per call" % (n, dt / n * 1e3))
exp, _ = orc.match_pair(ims[3], ims[9]); assert np.array_equal(m.match_pair(ims[3], ims[9]), exp)
print("ok")

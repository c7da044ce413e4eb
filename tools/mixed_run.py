"""tools/mixed_run.py -- the mixed-magnitudes grid of bench.py alone (sift128, image 50 x 2^30), for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from reconstructor_amd import synth
from reconstructor_amd.matcher import HipL2Matcher, all_pairs
n3, K3 = 100, 1500
pool3 = synth.world_pool("sift", 4 * K3, seed=1234)
x = np.stack([synth.image_descriptors("sift", i, K3, pool3, seed=1234) for i in range(n3)])
x[50] *= np.float32(2.0 ** 30)
xd = torch.from_numpy(x).cuda()
m = HipL2Matcher(device=0)
m.upload_batch_device(0, n3, xd.data_ptr(), K3, 128)
pr = all_pairs(n3)
o = torch.empty((len(pr), K3), dtype=torch.int32, device="cuda")
c = torch.empty((len(pr),), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(4):
    m.match_grid_device(pr, o.data_ptr(), K3, c.data_ptr())
m.ctx.check(m.ctx.lib.rcn_synchronize(m.ctx.h))
t0 = time.perf_counter()
for _ in range(5):
    m.match_grid_device(pr, o.data_ptr(), K3, c.data_ptr())
m.ctx.check(m.ctx.lib.rcn_synchronize(m.ctx.h))
print("%.3f ms per grid" % (1e3 * (time.perf_counter() - t0) / 5), m.stats())

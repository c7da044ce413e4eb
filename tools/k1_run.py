"""tools/k1_run.py [n_images K D reps] -- bare grid calls (no RCCL, no host lists) for profiling K1 under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from reconstructor_amd import synth
from reconstructor_amd.matcher import HipL2Matcher, all_pairs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
kind = {256: "superpoint", 128: "sift", 32: "orb"}[D]
ims = np.stack(synth.descriptor_set(kind, n, K, n_world=4 * K, seed=1234))
dev = torch.from_numpy(ims).cuda()
m = HipL2Matcher(device=0)
m.upload_batch_device(0, n, dev.data_ptr(), K, D)
pairs = all_pairs(n)
out = torch.empty((len(pairs), K), dtype=torch.int32, device="cuda")
cnt = torch.empty((len(pairs),), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
m.profile(True)
for _ in range(reps):
    m.match_grid_device(pairs, out.data_ptr(), K, cnt.data_ptr())
m.ctx.check(m.ctx.lib.rcn_synchronize(m.ctx.h))
st = m.stats()
calls, launches = max(1, st["profiled_calls"]), max(1, st["coarse_launches"])
pd = float(st["pair_distances"]) * calls / launches          # pair-distances of one launch (a large grid runs in pipeline chunks: several launches per call)
ms = st["coarse_ms"] / launches
print("K1 %.3f ms per launch, %.1f TFLOP/s, frac %.4f; matches %d; %d launches per call" % (ms, 2.0 * D * pd / ms * 1e-9, 2.0 * D * pd / ms * 1e-9 / 2500.0, int(cnt.sum().item()), launches // calls))

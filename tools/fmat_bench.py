"""tools/fmat_bench.py -- epipolar filter on a cfg-2-sized grid (4950 pairs x ~512 matches): time per grid, oracle check on a sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from reconstructor_amd import _lib, synth_fmat
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4950
ctx = _lib.Context(0)
sizes = np.random.default_rng(0).integers(400, 620, P)
t0 = time.time(); off, a, b = synth_fmat.grid(sizes, frac, seed=3); print("gen %.1fs" % (time.time() - t0))
d = {k: torch.from_numpy(v).cuda() for k, v in (("off", off), ("a", a), ("b", b))}
mask = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(P, dtype=torch.int32, device="cuda"); it = torch.zeros(P, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
def run():
    ctx.check(ctx.lib.rcn_fmat_filter_grid_device(ctx.h, P, d["off"].data_ptr(), d["a"].data_ptr(), d["b"].data_ptr(), mask.data_ptr(), cnt.data_ptr(), it.data_ptr(), None))
run(); ctx.check(ctx.lib.rcn_synchronize(ctx.h))
t0 = time.perf_counter()
for _ in range(5): run()
ctx.check(ctx.lib.rcn_synchronize(ctx.h))
dt = (time.perf_counter() - t0) / 5
its = it.cpu().numpy()
print("outliers %.0f%%: %.2f ms per grid of %d pairs (%d points); iterations mean %.1f max %d; inliers %.1f%%" % (100 * frac, 1e3 * dt, P, off[-1], its.mean(), its.max(), 100.0 * mask.sum().item() / off[-1]))

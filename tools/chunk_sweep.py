"""tools/chunk_sweep.py [n_images K] -- the grid call at several pipeline-chunk sizes (diagnostic build, RCN_CHUNK_ROWS): wall time per call,
K1 time summed over the chunk launches, exact-stage time, device memory held by the library."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RCN_LIB", os.path.join(ROOT, "tools", "librcn_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from reconstructor_amd import _lib, synth
from reconstructor_amd.matcher import HipL2Matcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D = 256
pool = synth.world_pool("superpoint", 4 * K, seed=1234)
dev = torch.empty((n, K, D), dtype=torch.float32, device="cuda")
for i in range(n):
    dev[i].copy_(torch.from_numpy(synth.image_descriptors("superpoint", i, K, pool, seed=1234)))
P = n * (n - 1) // 2
out = torch.empty((P, K), dtype=torch.int32, device="cuda")
cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for rows in sys.argv[3:] or ["67108864", "134217728", "268435456", "4294967296"]:
    os.environ["RCN_CHUNK_ROWS"] = rows
    free0, _ = torch.cuda.mem_get_info()
    m = HipL2Matcher(ctx=_lib.Context(0))
    m.upload_batch_device(0, n, dev.data_ptr(), K, D)
    sync = lambda: m.ctx.check(m.ctx.lib.rcn_synchronize(m.ctx.h))
    m.ctx.check(m.ctx.lib.rcn_match_grid_device(m.ctx.h, None, P, 0.7, out.data_ptr(), K, cnt.data_ptr()))
    sync()
    m.stats(); m.profile(True)
    t0 = time.perf_counter()
    for _ in range(2):
        m.ctx.check(m.ctx.lib.rcn_match_grid_device(m.ctx.h, None, P, 0.7, out.data_ptr(), K, cnt.data_ptr()))
    sync()
    dt = (time.perf_counter() - t0) / 2
    st = m.stats()
    free1, _ = torch.cuda.mem_get_info()
    print("chunk_rows %11s: %3d chunks, %8.1f ms per call, K1 %8.1f ms (%6.2f per launch), exact stages %6.1f ms, uniqueness %5.1f ms, library holds %.2f GB, matches %d"
          % (rows, st["chunks"], 1e3 * dt, st["coarse_ms"] / 2, st["coarse_ms"] / max(1, st["coarse_launches"]), st["rerank_ms"] / 2, st["unique_ms"] / 2,
             (free0 - free1) / 1e9, int(cnt.to(torch.int64).sum().item())), flush=True)
    m.clear()
    m.ctx.close()

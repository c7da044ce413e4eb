"""tools/w4_stress3.py -- diag lib: compare the coarse candidate table of a good and a bad step."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from reconstructor_amd import synth, pairgrid, _lib

n, K, steps = 100, 2048, 6
ims = np.stack(synth.descriptor_set("superpoint", n, K, n_world=4 * K, seed=1234))
dev = torch.from_numpy(ims).cuda()
ctx = _lib.Context(0)
lib = ctx.lib
lib.rcn_diag_coarse_table.restype = C.c_int
lib.rcn_diag_coarse_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]
sh = pairgrid.Shard(ctx, 0, 1, pairgrid.unique_id())
sh.reserve(n, K, 256)
P = n * (n - 1) // 2
out = torch.empty((P, K), dtype=torch.int32, device="cuda")
cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
tabs = []
for it in range(steps):
    sh.exchange(dev.data_ptr())
    if "--sync" in sys.argv:
        ctx.check(lib.rcn_synchronize(ctx.h)); torch.cuda.synchronize()
    sh.match(0.7, out.data_ptr(), K, cnt.data_ptr())
    ctx.check(lib.rcn_synchronize(ctx.h))
    model = (C.c_double * 8)()
    lib.rcn_diag_coarse_table(ctx.h, None, 0, model)
    kq, mask = int(model[4]), int(model[3])
    cand = np.zeros((P, kq, 2), np.uint32)
    ctx.check(lib.rcn_diag_coarse_table(ctx.h, cand.ctypes.data, cand.size, model))
    tabs.append(cand[:, :K].copy())
    print("step", it, "matches", int(cnt.sum().item()))
ref = tabs[0]
for it in range(1, steps):
    d = np.argwhere((tabs[it] != ref).any(2))
    if len(d) == 0:
        continue
    print("step", it, ":", len(d), "(pair, query) entries differ; pairs", sorted(set(d[:, 0]))[:6])
    p, q = d[0]
    sel = d[d[:, 0] == p][:, 1]
    print("  pair", p, ":", len(sel), "queries differ; query blocks of 32:", np.bincount(sel // 32, minlength=64).tolist())
    for q in sel[:6]:
        a, b = ref[p, q], tabs[it][p, q]
        print("   q", q, "good best/second idx", a[0] & mask, a[1] & mask, "val", hex(a[0] & ~np.uint32(mask)), hex(a[1] & ~np.uint32(mask)),
              "| bad idx", b[0] & mask, b[1] & mask, "val", hex(b[0] & ~np.uint32(mask)), hex(b[1] & ~np.uint32(mask)))
    # which train rows are involved: good candidates that the bad run lost
    lost = np.concatenate([ref[p, sel, 0] & mask, ref[p, sel, 1] & mask])
    print("  train tiles (64 rows) of the good candidates at the differing queries:", np.bincount(lost // 64, minlength=K // 64).tolist())
    break

"""tools/w4_stress2.py -- the bench's exact step (shard exchange + match) repeated; run-to-run comparison of the tables."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from reconstructor_amd import synth, pairgrid, _lib
from reconstructor_amd.matcher import HipL2Matcher

n, K, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ims = np.stack(synth.descriptor_set("superpoint", n, K, n_world=4 * K, seed=1234))
dev = torch.from_numpy(ims).cuda()
ctx = _lib.Context(0)
sh = pairgrid.Shard(ctx, 0, 1, pairgrid.unique_id())
sh.reserve(n, K, 256)
P = n * (n - 1) // 2
outs = [torch.empty((P, K), dtype=torch.int32, device="cuda") for _ in range(2)]
cnts = [torch.empty((P,), dtype=torch.int32, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
ref = None
for it in range(steps):
    o, c = outs[it & 1], cnts[it & 1]
    sh.exchange(dev.data_ptr())
    sh.match(0.7, o.data_ptr(), K, c.data_ptr())
    if "--sync" in sys.argv:
        ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    if it >= 1:
        ctx.check(ctx.lib.rcn_synchronize(ctx.h))
        if ref is None:
            ref = outs[0].clone(); refc = cnts[0].clone()
            print("step 0 matches", int(refc.sum().item()))
        same = torch.equal(o, ref)
        if not same:
            d = (o != ref).nonzero()
            dn = d.cpu().numpy()
            q = dn[:, 1]
            print("step", it, "differs from step 0 in", len(d), "entries; matches", int(c.sum().item()))
            print("   by 32-query block of the 512-tile:", np.bincount((q % 512) // 32, minlength=16).tolist())
            print("   by query tile:", np.bincount(q // 512, minlength=4).tolist(), " by lane r:", np.bincount(q % 32, minlength=32).tolist())
            # position of the pair within its run of consecutive pairs sharing the query image (groups of <= 4)
            pr = pairgrid.shard_pairs(n, 1, 0)
            start = np.r_[0, np.flatnonzero(np.diff(pr[:, 0])) + 1]
            pos = np.arange(len(pr)) - np.repeat(start, np.diff(np.r_[start, len(pr)]))
            print("   by position in run %4:", np.bincount(pos[dn[:, 0]] % 4, minlength=4).tolist(), " train image - query image:", np.bincount(np.minimum(pr[dn[:,0],1]-pr[dn[:,0],0], 9), minlength=10).tolist())
            newv = o[d[:8, 0], d[:8, 1]].tolist(); oldv = ref[d[:8, 0], d[:8, 1]].tolist()
            print("   sample new/old:", list(zip(newv, oldv)))
print("done")

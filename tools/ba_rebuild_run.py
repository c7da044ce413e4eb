"""tools/ba_rebuild_run.py -- solve times of four BA sizes when the pair lists of the Schur build are BUILT by every timed solve (a solve of
another graph in front of each drops the kept lists): what the reference's loop, which adds a view before every adjust, would see.
tools/ba_run.py re-solves one scene and reuses them.  RCN_LIB selects the library."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from reconstructor_amd import _lib, ba, synth_ba
ctx = _lib.Context(0)
for nc, npts in ((3, 300), (25, 1500), (200, 20000), (1000, 100000)):
    a = synth_ba.make_scene(nc, npts, seed=2024)
    b = synth_ba.make_scene(3, 60, obs_per_point=3, seed=7)
    ts = []
    for i in range(7):
        ba.solve_scene(ctx, b)
        P, I, X, s = ba.solve_scene(ctx, a)
        assert s["pair_lists_reused"] == 0
        if i >= 2: ts.append(s["solve_seconds"] * 1e3)
    ts.sort()
    print(nc, "cams: solve ms (lists built every time) median %.4f min %.4f, %d iterations" % (ts[len(ts)//2], ts[0], s["iterations"]))

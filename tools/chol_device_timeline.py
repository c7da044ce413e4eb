#!/usr/bin/env python3
"""tools/chol_device_timeline.py [n_cams n_points] -- the factorisation's kernels on the DEVICE clock (diagnostic build: every kernel
of the three streams stamps wall_clock64 when its first workgroup enters, when its gate lets it pass and when its last
workgroup leaves).  Unlike a rocprofv3 kernel trace this does not stretch dependent launches or cross-stream hand-offs.
Prints, per block step of the LAST factorisation of the solve, microseconds relative to the start of the step's diagonal kernel:
kind[start gate-passed end].  B = bulk update (for a two-panel kernel the middle figure is the time its last head tile finished)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RCN_LIB", os.path.join(ROOT, "tools", "librcn_diag.so"))
sys.path.insert(0, ROOT)
import ctypes as C

import torch  # noqa: E402

from reconstructor_amd import _lib, ba, synth_ba  # noqa: E402


def main():
    nc, npts = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 100000)
    ctx = _lib.Context(0)
    sc = synth_ba.make_scene(nc, npts, seed=2024)
    P, I, X, s = ba.solve_scene(ctx, sc)
    nblk = (s["reduced_dim"] + 1 + 127) // 128
    buf = torch.zeros(3 * 8 * (nblk + 2), dtype=torch.int64, device="cuda")
    fn = ctx.lib.rcn_diag_timeline_set
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(buf.data_ptr()) == 0
    P, I, X, s = ba.solve_scene(ctx, sc)
    torch.cuda.synchronize()
    assert fn(None) == 0
    t = buf.cpu().numpy().reshape(nblk + 2, 8, 3).astype("float64") / 100.0     # 100 MHz -> us
    # (at the LAST step of a two-level super-step, which has no critical tile of its own: P / C / C2 = the head rows' column 0, their
    #  product with the super-block's inverse, the update of the next super-diagonal block; T1 / T2 = column 0 and the product for all rows
    #  below; W = a block row of the super-block's inverse)
    names = {0: "D", 1: "T1", 2: "T2", 3: "P", 4: "C", 5: "C2", 6: "B", 7: "W"}
    print("%d iterations, chol %.3f ms per iteration, %d block steps" % (s["iterations"], 1e3 * s["cholesky_seconds"] / s["iterations"], nblk))
    prev = None
    for k in range(nblk):
        d0 = t[k, 0, 0]
        if d0 == 0:
            continue
        parts = []
        for kind in range(8):
            a, g, e = t[k, kind]
            if a == 0 and e == 0:
                continue
            parts.append("%s[%.0f %.0f %.0f]" % (names[kind], a - d0, (g - d0) if g else float("nan"), e - d0))
        step = (d0 - prev) if prev else 0.0
        prev = d0
        print("step %2d  (+%6.1f)  %s" % (k, step, "  ".join(parts)))
    print("whole factorisation on the device clock: %.3f ms" % ((t[nblk - 1, 0, 2] - t[0, 0, 0]) / 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/mfma_error_model.py <out.json> -- measures the coarse pass's error model ON THE HARDWARE.

DESIGN.md section 5 bounds |MFMA accumulator - exact value| by eps(q); every certificate of the matcher rests on
it.  The diagnostic build (tools/librcn_diag.so, -DRCN_DIAG) returns the packed (best, second) table of the
last grid call (rcn_diag_coarse_table); for every query row this script computes the EXACT accumulator values
A* = s^2(|t|^2/2 - q.t) + BIAS of ALL train rows in float64 and checks
    both candidates:   acc in [trunc, trunc + quantum)  and  |acc - A*| <= eps   =>  A* in [trunc - eps, trunc + quantum + eps]
    every other row:   acc >= trunc(second)                                       =>  A* >= trunc(second) - eps
and reports, per data set, the largest violation-free margin: max over rows of the distance of A* outside the
quantisation interval, in units of eps (0 = inside the interval; 1 = at the bound).
Run with RCN_LIB=tools/librcn_diag.so (tests/test_mfma_error_model_gpu.py does)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def data_sets():
    from reconstructor_amd import synth
    out = {}
    out["cfg2 pair (2048 x 2048 x 256 SuperPoint-like)"] = synth.descriptor_set("superpoint", 2, 2048, n_world=8192, seed=1234)
    out["sift 128-d (1500 x 1700)"] = synth.descriptor_set("sift", 2, [1500, 1700], n_world=4000, seed=3)
    out["orb-as-float 32-d (1200 x 900)"] = synth.descriptor_set("orb", 2, [1200, 900], n_world=2500, seed=4)
    rng = np.random.default_rng(8)
    base = rng.standard_normal((40, 256)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    t = np.repeat(base, 16, axis=0) + (rng.standard_normal((640, 256)) * 1e-5).astype(np.float32)
    q = base + (rng.standard_normal((40, 256)) * 1e-5).astype(np.float32)
    out["adversarial near-duplicates (40 x 640 x 256)"] = [q, t]
    big = synth.descriptor_set("superpoint", 2, 1024, n_world=2048, seed=9)
    out["SuperPoint-like scaled by 2^40 (1024 x 1024)"] = [(x * np.float32(2.0 ** 40)).astype(np.float32) for x in big]
    return out


def measure(m, lib, q, t):
    m.clear()
    m.upload(0, q)
    m.upload(1, t)
    m.match_grid(np.array([[0, 1]], np.int32), len(q))
    model = (C.c_double * 8)()
    m.ctx.check(lib.rcn_diag_coarse_table(m.ctx.h, None, 0, model))
    s, bias, nmax, mask, kq_stride, DP, hn_max, n_pairs = [model[i] for i in range(8)]
    mask, kq_stride = int(mask), int(kq_stride)
    cand = np.zeros((kq_stride, 2), np.uint32)
    m.ctx.check(lib.rcn_diag_coarse_table(m.ctx.h, cand.ctypes.data, cand.size, model))
    assert m.stats()["used_mfma_path"] == 1
    K1, K2 = len(q), len(t)
    cand = cand[:K1]
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    A = 0.5 * s * s * (t64 ** 2).sum(1)[None, :] + bias - s * s * (q64 @ t64.T)          # exact accumulators, K1 x K2
    u = 2.0 ** -11
    nq = np.sqrt((q64 ** 2).sum(1)) * (1 + 1e-12)
    eps = ((2 * u + u * u) * s * s * nq * nmax + 2.0 ** -14 * np.sqrt(DP) * s * (nq + nmax) + 1e-9 +
           (DP + 8) * 2.0 ** -23 * (hn_max + s * s * nq * nmax) + 6.0e-8 * hn_max)
    inv = np.uint32(~np.uint32(mask) & np.uint32(0xFFFFFFFF))
    lo = (cand & inv).view(np.float32).astype(np.float64)                                   # trunc
    hi = ((cand & inv) + np.uint32(mask + 1)).view(np.float32).astype(np.float64)           # trunc + quantum
    idx = (cand & np.uint32(mask)).astype(np.int64)
    rows = np.arange(K1)
    worst = 0.0
    for c in range(2):
        a = A[rows, idx[:, c]]
        out = np.maximum(np.maximum(lo[:, c] - a, a - hi[:, c]), 0.0) / eps
        assert (out <= 1.0).all(), ("candidate outside the bound", c, float(out.max()))
        worst = max(worst, float(out.max()))
    nc = A.copy()
    nc[rows, idx[:, 0]] = np.inf
    nc[rows, idx[:, 1]] = np.inf
    short = np.maximum(lo[:, 1][:, None] - nc, 0.0) / eps[:, None]                          # how far below trunc(second)
    assert (short <= 1.0).all(), ("a non-candidate row beats the second candidate by more than eps", float(short.max()))
    worst_nc = float(short.max())
    # the exact top-2 of every row is among what the bound allows: sanity of the proposal itself
    quantum = float((hi - lo).max())
    return {"K1": K1, "K2": K2, "DP": int(DP), "scale_log2": float(np.log2(s)), "eps_max": float(eps.max()), "eps_min": float(eps.min()),
            "quantum_max": quantum, "candidate_excess_over_quantisation_interval_in_eps": worst,
            "non_candidate_shortfall_in_eps": worst_nc, "rows": int(K1), "pair_distances_checked": int(K1) * int(K2)}


def main():
    import torch  # noqa: F401
    from reconstructor_amd import _lib
    from reconstructor_amd.matcher import HipL2Matcher
    lib = _lib.load()
    assert b"DIAGNOSTIC" in lib.rcn_version(), "run with RCN_LIB=tools/librcn_diag.so"
    lib.rcn_diag_coarse_table.restype = C.c_int
    lib.rcn_diag_coarse_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]
    m = HipL2Matcher(device=0)
    res = {}
    for name, (q, t) in data_sets().items():
        res[name] = measure(m, lib, np.ascontiguousarray(q), np.ascontiguousarray(t))
        print(name, json.dumps(res[name]), flush=True)
    res["_note"] = ("units of eps(q): 0 = the exact accumulator lies inside the truncated candidate's quantisation interval, "
                    "1 = at the certified bound; measured on MI355X with v_mfma_f32_32x32x16_f16")
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()

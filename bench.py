#!/usr/bin/env python3
"""bench.py -- headline benchmark: descriptor pair-distances/s of the all-pairs matcher.

Workload at EVERY N (strong scaling): BASELINE.json configs[2], the configuration the metric's
"@1/2/4/8 GPU" is quoted on and which fits one GPU -- synthetic 1000 images x 4096 SuperPoint-like
256-d keypoints, exact brute-force L2 2-NN + ratio + uniqueness over all 499 500 image pairs.
One "step" = one pass of the hot path over the whole grid, through the sharded-grid C ABI
(rcn_shard_*, csrc/shard.hip) at every N including 1:
    exchange   each rank's block of fp32 descriptors is resident in HBM when the step starts; row
               statistics, RCCL all-reduce of the scale statistics, fp16 conversion of the local
               block, in-place RCCL all-gather of the fp16 payload (+ fp32 rows on a side stream)
    match      this rank's share of the canonical grid (pair number p -> rank p % N)
    materialise  (query, train) lists compacted on the GPU and copied to pinned host memory
               ("incl. result materialisation on host"); the copy of step k overlaps step k+1
The N = 1 line also carries configs[1] (100 x 2048 x 256, 4950 pairs) under "cfg2", the CPU baseline,
the BA legs (cfg 4 / cfg 5; BA does not shard: "replicas only") and the two widened rows.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (a child
`python -m torch.distributed.run`, before this process has touched the GPU); fewer than N visible devices is an
error (exit 2) -- it never falls back to a smaller world.

Control plane (rendezvous id, barriers, max-over-ranks time): torch.distributed / gloo.  Data plane:
RCCL called directly from librcn.so.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 256
WORKLOADS = {"cfg3": (1000, 4096), "cfg2": (100, 2048)}     # BASELINE.json configs[2] / configs[1]: images, keypoints per image
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense bf16/f16 MFMA peak (same guide)


def host_threads():
    """Threads for the CPU baselines: the GPU box's CPU share for one GPU is 16 cores even though
    the affinity mask lists every core of the host (a cap, not nproc: both are in the line)."""
    return max(1, min(16, len(os.sched_getaffinity(0))))


def host_info():
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "threads_used_cap": 16}


def source_hash():
    """Identity of the binary: sha256 over the sources librcn.so is built from.  Profile summaries under
    profiles/ carry the hash they were taken at; a counter-derived figure is quoted only when it matches."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.listdir(os.path.join(ROOT, "reconstructor_amd", "csrc")))
    for f in files:
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(ROOT, "reconstructor_amd", "csrc", f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rcn.h"), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes -- only if that
    profile was taken on THIS source (else None: a stale file is not a measurement of this run)."""
    path = os.path.join(ROOT, "profiles", "r05_match_traffic.json")
    if not os.path.exists(path):
        return None, "no PMC profile committed for this round"
    d = json.load(open(path))
    if d.get("source_hash") != source_hash():
        return None, "profiles/r05_match_traffic.json was taken at source %s, this binary is %s: not quoted" % (d.get("source_hash"), source_hash())
    e = d.get(kernel_key)
    if not e:
        return None, "no entry for " + kernel_key
    return e["traffic_bytes_per_launch"], e.get("note", "")


def cpu_baseline(images, n_threads):
    """Oracle (CPU restatement, kind "port"; rebuilt -march=native on this box) on a bounded sample of the
    same workload: about ten seconds at the box's cores for this GPU, plus the reference's own thread
    count (MAX_NUM_THREADS = 4, SequentialReconstructor.h:17) on a smaller sample."""
    from oracle import orc
    orc.build(native=True)
    n = len(images)
    allp = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], np.int32)
    rng = np.random.default_rng(0)
    K = images[0].shape[0]
    scale = (2048 * 2048) / float(K * K)        # pairs of the sample shrink with the pair size

    def timed(n_pairs, threads):
        pick = allp[rng.choice(len(allp), size=min(n_pairs, len(allp)), replace=False)]
        t0 = time.perf_counter()
        orc.match_grid(images, pick, threads=threads)
        dt = time.perf_counter() - t0
        return sum(images[a].shape[0] * images[b].shape[0] for a, b in pick) / dt, len(pick), dt

    orc.match_grid(images, allp[:n_threads], threads=n_threads)  # warm (page-in, transposes)
    v, n, dt = timed(max(n_threads, int(100 * n_threads * scale)), n_threads)
    v4, n4, dt4 = timed(max(4, int(96 * scale)), 4)
    # every core the affinity mask lists (OpenMP threads of this one process).  The box gives a one-GPU job a CPU
    # share of 16 cores, so the figure shows what that share yields with every core asked for, not 256 free cores; stated as measured
    n_all = max(1, min(256, len(os.sched_getaffinity(0))))
    all_cores = None
    if n_all > n_threads:
        va, na, dta = timed(max(n_all, int(8 * n_all * scale)), n_all)
        all_cores = {"value": va, "cores": n_all, "sample": "%d pairs, %.1f s" % (na, dta),
                     "note": "threads = cores in the affinity mask; the pool's CPU share for a one-GPU box is 16 cores"}
    return {"value": v, "unit": "pair-distances/s", "cores": n_threads, "kind": "port", "at_all_affinity_cores": all_cores,
            "sample": "%d image pairs (%dx%dx256 each) among the first %d images of the workload, oracle/match_oracle.c (-O3 -march=native), OpenMP over pairs, %.1f s"
                      % (n, K, K, len(images), dt),
            "host": host_info(),
            "at_reference_thread_count": {"value": v4, "cores": 4, "sample": "%d pairs, %.1f s" % (n4, dt4)}}


def ba_leg(ctx, with_cpu):
    """BA LM-iterations/s on cfg 4 / cfg 5 (single GPU: BA does not shard, "replicas only"),
    with the oracle's cfg-4 solve as the CPU baseline."""
    from reconstructor_amd import ba_bench
    sc4, out = ba_bench.run(ctx)
    if with_cpu:
        from oracle import orc_ba
        threads = host_threads()
        t0 = time.perf_counter()
        P, I, X, s = orc_ba.solve(sc4, threads=threads)
        out["cpu_baseline"] = {"value": s["iterations"] / s["solve_seconds"], "unit": "LM-iterations/s",
                               "cores": threads, "kind": "port",
                               "sample": "cfg4 full solve (%d iterations, %.2f s), oracle/ba_oracle.c" % (s["iterations"], time.perf_counter() - t0),
                               "final_rms_px": s["final_rms_px"]}
        out["cfg4"]["rms_diff_vs_cpu_px"] = abs(out["cfg4"]["final_rms_px"] - s["final_rms_px"])
        t0 = time.perf_counter()
        P, I, X, s4 = orc_ba.solve(sc4, threads=4)      # options.num_threads = 4 in the reference (BundleAdjuster.cpp:134)
        out["cpu_baseline"]["at_reference_thread_count"] = {"value": s4["iterations"] / s4["solve_seconds"], "cores": 4,
                                                            "sample": "cfg4 full solve, %.2f s" % (time.perf_counter() - t0)}
    # cfg 5 against the CPU oracle's own solve of the same scene (fixture written once in the build container:
    # tests/golden/make_ba_cfg5_golden.py; the oracle needs minutes for it, too long for the bench run)
    gpath = os.path.join(ROOT, "tests", "golden", "ba_cfg5.npz")
    if "cfg5" in out and os.path.exists(gpath):
        g = np.load(gpath)
        out["cfg5"]["rms_diff_vs_cpu_px"] = abs(out["cfg5"]["final_rms_px"] - float(g["final_rms_px"]))
        out["cfg5"]["iterations_equal_to_cpu"] = bool(out["cfg5"]["lm_iterations"] == int(g["iterations"]))
        out["cfg5"]["cpu_oracle"] = {"final_rms_px": float(g["final_rms_px"]), "iterations": int(g["iterations"]),
                                     "seconds": float(g["oracle_seconds"]), "threads": int(g["oracle_threads"]),
                                     "lm_iterations_per_s": int(g["iterations"]) / float(g["oracle_seconds"]),
                                     "source": "tests/golden/ba_cfg5.npz (oracle/ba_oracle.c in the build container)"}
    out["landmark_sweep"] = sweep_leg(ctx, with_cpu)
    return out


def fmat_leg(ctx, with_cpu):
    """Epipolar filter (GeometricFilter::estimateFundamental inside the pair loop) on a
    cfg-2-sized grid: 4950 pairs x 400..620 matches, 30 % gross outliers, inputs resident in HBM
    (rcn_fmat_filter_grid_device), timed over 5 back-to-back grids."""
    import ctypes as C
    import torch
    from reconstructor_amd import synth_fmat
    P = 4950
    sizes = np.random.default_rng(0).integers(400, 620, P)
    off, a, b = synth_fmat.grid(sizes, 0.3, seed=3)
    d = {k: torch.from_numpy(v).cuda() for k, v in (("off", off), ("a", a), ("b", b))}
    mask = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(P, dtype=torch.int32, device="cuda")
    it = torch.zeros(P, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def run():
        ctx.check(ctx.lib.rcn_fmat_filter_grid_device(ctx.h, P, d["off"].data_ptr(), d["a"].data_ptr(), d["b"].data_ptr(),
                                                      mask.data_ptr(), cnt.data_ptr(), it.data_ptr(), None))
    run()
    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    dt = (time.perf_counter() - t0) / 5
    its = it.cpu().numpy().astype(np.int64)
    n = np.diff(off).astype(np.int64)
    # the reference's own work: every executed iteration scores its (on average ~2) matrices against all n points,
    # 44 flop per (matrix, point) -- 20 mul + 18 add + 2 div + 4 for the squares and the scale; the 7-point solves add < 2 %
    evals = float((its * n).sum()) * 2.0
    out = {"workload": "4950 pairs x 400..620 matches (%d points), 30%% outliers" % off[-1], "pairs_per_s": P / dt,
           "ms_per_grid": 1e3 * dt, "mean_iterations": float(its.mean()), "inlier_share": float(mask.sum().item()) / float(off[-1]),
           "roofline": {"bound": "fp64 valu", "achieved": evals * 44.0 / dt * 1e-12, "peak": 78.6, "unit": "TFLOP/s",
                        "frac": evals * 44.0 / dt * 1e-12 / 78.6, "traffic": None,
                        "note": "algorithmic flop = executed iterations x 2 matrices x n points x 44 (unfused: the reference's arithmetic has no FMA, so the usable peak is half of 78.6)"}}
    if with_cpu:
        from oracle import orc_fmat
        sel = P
        orc_fmat.filter_grid(off[:65], a[:off[64]], b[:off[64]], threads=host_threads())       # warm
        t0 = time.perf_counter()
        for _ in range(5):
            m0, c0, i0 = orc_fmat.filter_grid(off[:sel + 1], a[:off[sel]], b[:off[sel]], threads=host_threads())
        dtc = (time.perf_counter() - t0) / 5
        out["cpu_baseline"] = {"value": sel / dtc, "unit": "pairs/s", "cores": host_threads(), "kind": "port",
                               "sample": "the whole grid 5 times, oracle/fmat_oracle.c, OpenMP over pairs, %.2f s" % (5 * dtc)}
        out["equal_to_cpu"] = bool((mask[:off[sel]].cpu().numpy().astype(bool) == m0).all() and (cnt[:sel].cpu().numpy() == c0).all())
    return out


def sweep_leg(ctx, with_cpu):
    """Landmark validity sweep (checkLandmarkValidity) on the cfg-5 observation
    graph: 1000 cameras, 100k landmarks, ragged tracks of up to 10 observations, inputs resident in
    HBM (rcn_landmark_validity_device), timed over 20 back-to-back sweeps."""
    import ctypes as C
    import torch
    from reconstructor_amd import _lib, synth_ba
    c = synth_ba.make_validity_case(1000, 100000, obs_per_point=10, seed=9, defect_rate=0.1)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in c.items()}
    n_pts, n_obs = len(c["points"]), len(c["obs_cam"])
    inl = torch.zeros(n_pts, dtype=torch.uint8, device="cuda")
    keep = torch.zeros(n_obs, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    pb = _lib.LandmarkProblem(1000, n_pts, n_obs, 0, dev["poses34"].data_ptr(), dev["intrinsics"].data_ptr(), dev["points"].data_ptr(),
                              dev["pt_off"].data_ptr(), dev["obs_cam"].data_ptr(), dev["obs_xy"].data_ptr())
    torch.cuda.synchronize()

    def sweep():
        ctx.check(ctx.lib.rcn_landmark_validity_device(ctx.h, C.byref(pb), 4.0, 1.0, inl.data_ptr(), keep.data_ptr(), cnt.data_ptr()))
    for _ in range(3):
        sweep()
    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        sweep()
    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    dt = (time.perf_counter() - t0) / reps
    # streamed once per sweep: obs_cam 4 + obs_xy 8 + keep 1 B per observation, point 24 + offset 4 + flag 1 B per
    # landmark (poses / intrinsics / centres, 168 B per camera, stay in L2)
    alg = 13.0 * n_obs + 29.0 * n_pts
    out = {"workload": "1000 cams / 100000 landmarks / %d observations" % n_obs, "landmarks_per_s": n_pts / dt,
           "observations_per_s": n_obs / dt, "ms_per_sweep": 1e3 * dt, "valid_landmarks": int(cnt.item()),
           "roofline": {"bound": "hbm", "achieved": alg / dt * 1e-9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / dt * 1e-9 / 8000.0, "traffic": None}}
    if with_cpu:
        from oracle import orc_validity
        t0 = time.perf_counter()
        for _ in range(5):
            inl0, keep0 = orc_validity.landmark_validity(**c)
        dtc = (time.perf_counter() - t0) / 5
        out["cpu_baseline"] = {"value": n_pts / dtc, "unit": "landmarks/s", "cores": 1, "kind": "port",
                               "sample": "5 full sweeps, oracle/validity_oracle.c, %.2f s" % (5 * dtc)}
        out["equal_to_cpu"] = bool((inl.cpu().numpy().astype(bool) == inl0).all() and (keep.cpu().numpy().astype(bool) == keep0).all())
    return out


def cfg1_leg(ctx, with_cpu):
    """BASELINE.json configs[0], the regime the reference actually runs in (README.md:50-53, SequentialReconstructor.cpp:978-1103):
    25 images x ~1500 keypoints through the whole pair loop -- match, epipolar filter, host lists -- for SIFT-like 128-d and
    ORB-as-float 32-d descriptors, then the incremental loop's 23 global bundle adjustments (3 -> 25 cameras) through the
    device-resident session with the validity sweeps around every solve.  The Fountain JPEGs cannot be turned into
    descriptors here (no OpenCV): a synthetic scene of the same shape stands in.  GPU beside the CPU oracle at the
    reference's own 4 threads (MAX_NUM_THREADS, options.num_threads), same box, same inputs, results compared."""
    import ctypes as C
    import torch
    from reconstructor_amd import ba, fmat, synth, synth_ba
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    out = {"workload": "25 images x 1400..1592 keypoints (synthetic scene; the Fountain set needs OpenCV), 300 image pairs; then 23 global BAs, 3 -> 25 cameras"}
    n = 25
    ks = [1400 + 8 * ((7 * i) % 25) for i in range(n)]
    pairs = all_pairs(n)
    P, stride = len(pairs), max(ks)
    m = HipL2Matcher(ctx=ctx)
    sync = lambda: ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    for kind, D in (("sift", 128), ("orb", 32)):
        ims, coords, _ = synth.scene_set(kind, n, ks, n_world=5000, seed=19)
        m.clear()
        t0 = time.perf_counter()
        for i in range(n):
            m.upload(i, ims[i])
            m.upload_coords(i, coords[i])
        sync()
        t_up_each = time.perf_counter() - t0
        # the batched host ingest (rcn_desc_upload_batch + rcn_coords_upload_batch: one synchronisation each), from pageable rows
        # as a detector's std::vector would be and from pinned rows (rcn_host_alloc); the second call of a shape reuses the block
        m.clear()
        m.upload_batch(0, ims); m.upload_coords_batch(0, coords); sync()
        t0 = time.perf_counter()
        m.upload_batch(0, ims)
        m.upload_coords_batch(0, coords)
        sync()
        t_up = time.perf_counter() - t0
        pin = C.c_void_p()
        nbytes = sum(im.nbytes for im in ims)
        ctx.check(ctx.lib.rcn_host_alloc(C.byref(pin), nbytes))
        pinned, o = [], 0
        for im in ims:
            v = np.ctypeslib.as_array((C.c_float * im.size).from_address(pin.value + o)).reshape(im.shape)
            v[...] = im
            pinned.append(v)
            o += im.nbytes
        m.upload_batch(0, pinned); sync()
        t0 = time.perf_counter()
        m.upload_batch(0, pinned)
        m.upload_coords_batch(0, coords)
        sync()
        t_up_pinned = time.perf_counter() - t0
        m.upload_batch(0, ims); sync()          # the pinned block goes away below: the resident rows are the pageable ones' copy
        ctx.lib.rcn_host_free(pin)
        tab = torch.empty((P, stride), dtype=torch.int32, device="cuda")
        cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
        ver = torch.empty((P,), dtype=torch.int32, device="cuda")
        offs = np.zeros(P + 1, np.int64)
        qt = np.zeros((P * stride, 2), np.int32)
        total = C.c_int64(0)
        torch.cuda.synchronize()

        def loop(split):
            ts = [time.perf_counter()]
            m.match_grid_device(pairs, tab.data_ptr(), stride, cnt.data_ptr())
            if split:
                sync(); ts.append(time.perf_counter())
            m.filter_table_device(pairs, tab.data_ptr(), stride, cnt.data_ptr(), ver.data_ptr())
            if split:
                sync(); ts.append(time.perf_counter())
            ctx.check(ctx.lib.rcn_match_compact_begin(ctx.h, C.c_void_p(tab.data_ptr()), stride, C.c_void_p(cnt.data_ptr()), P, offs.ctypes.data,
                                                      qt.ctypes.data, len(qt), C.byref(total)))
            ctx.check(ctx.lib.rcn_match_compact_wait(ctx.h))
            ts.append(time.perf_counter())
            return ts
        loop(False)                                   # warm (workspaces)
        ts = loop(True)
        t_all = min((lambda r: r[-1] - r[0])(loop(False)) for _ in range(3))
        g = {"upload_seconds": t_up, "upload_seconds_pinned_rows": t_up_pinned, "upload_seconds_one_call_per_image": t_up_each, "match_seconds": ts[1] - ts[0], "filter_seconds": ts[2] - ts[1], "lists_seconds": ts[3] - ts[2],
             "gpu_seconds": t_all, "matches_kept": int(total.value), "pairs_per_s": P / t_all,
             "note": "gpu_seconds = match + filter + host lists back to back (no synchronisation in between); the three stage times are one run with a synchronisation after each"}
        if with_cpu:
            from oracle import orc, orc_fmat
            orc.match_grid(ims[:2], np.array([[0, 1]], np.int32), threads=4)
            t0 = time.perf_counter()
            tab0, cnt0 = orc.match_grid(ims, pairs, threads=4)
            t1 = time.perf_counter()
            off, a, b = fmat.matches_to_csr(coords, pairs, tab0)
            t2 = time.perf_counter()
            mask, c0, _ = orc_fmat.filter_grid(off, a, b, threads=4)
            t3 = time.perf_counter()
            want = tab0.copy()
            for p, (i, j) in enumerate(pairs):
                q = np.flatnonzero(tab0[p, :ks[i]] >= 0)
                if len(q) >= 7:
                    want[p, q[~mask[off[p]:off[p + 1]]]] = -1
            got = tab.cpu().numpy()
            g.update({"cpu_seconds": (t1 - t0) + (t3 - t2), "cpu_match_seconds": t1 - t0, "cpu_filter_seconds": t3 - t2, "cpu_threads": 4,
                      "equal_to_cpu": bool(np.array_equal(got, want) and np.array_equal(ver.cpu().numpy(), c0) and int((want >= 0).sum()) == int(total.value))})
        out["pair_loop_%s%d" % (kind, D)] = g
        del tab, cnt, ver
    m.clear()

    # ---- the incremental loop: a global BA after every registered view, sweeps around it (SequentialReconstructor.cpp:1040-1094)
    sc = synth_ba.make_scene(25, 1500, obs_per_point=6, seed=31)
    xy_all = sc["obs_uv"].astype(np.int32)
    ses = ba.BaSession(ctx)
    gpu_s, cpu_s, gpu_sweep_s, cpu_sweep_s, steps, equal = 0.0, 0.0, 0.0, 0.0, [], True
    try:
        sid, live = {}, np.zeros(len(sc["points"]), bool)
        dead = np.zeros(len(sc["points"]), bool)           # landmarks the post-solve sweep removed: their later observations are dropped
        for ncam in range(1, n + 1):
            ses.add_camera(sc["poses"][ncam - 1], sc["intrinsics"][ncam - 1])
            keep = sc["obs_cam"] < ncam
            live_now = (np.bincount(sc["obs_pt"][keep], minlength=len(sc["points"])) >= 2) & ~dead
            new_pts = np.flatnonzero(live_now & ~live)
            o_new = np.flatnonzero((sc["obs_cam"] == ncam - 1) & live[sc["obs_pt"]] & ~dead[sc["obs_pt"]])
            ses.add_observations([sid[j] for j in sc["obs_pt"][o_new]], sc["obs_cam"][o_new], xy_all[o_new])
            if len(new_pts):
                first = ses.add_points(sc["points"][new_pts])
                for k, j in enumerate(new_pts):
                    sid[j] = first + k
                o_tr = np.flatnonzero(np.isin(sc["obs_pt"], new_pts) & (sc["obs_cam"] < ncam))
                ses.add_observations([sid[j] for j in sc["obs_pt"][o_tr]], sc["obs_cam"][o_tr], xy_all[o_tr])
            live = live_now
            if ncam < 3:
                continue
            poses, intr = ses.cameras()
            flat = None
            if with_cpu:
                pt, cam, xy = ses.graph()
                X = ses.points()
                flat = {"poses": poses, "intrinsics": intr, "points": X, "obs_uv": xy.astype(np.float64), "obs_cam": cam, "obs_pt": pt}
            p34 = synth_ba.poses_to_34(poses)
            # the sweep BEFORE the solve runs at full cost but with thresholds that erase nothing: the synthetic loop hands the
            # solver a perturbed estimate (6.6 px RMS) where the reference's addNextView hands it triangulated points, so the
            # reference's 4-px test would empty the graph here; the sweep AFTER the solve uses the reference's thresholds
            t0 = time.perf_counter()
            inl_a, er_a = ses.validity(p34, 1e9, 0.0)
            t1 = time.perf_counter()
            sg = ses.solve()
            t2 = time.perf_counter()
            p34b = synth_ba.poses_to_34(ses.cameras()[0])
            t3 = time.perf_counter()
            inl_b, er_b = ses.validity(p34b, 4.0, 1.0)
            new_idx, removed = ses.remove_outliers()
            t4 = time.perf_counter()
            if removed:                                    # landmarks compacted on the device: follow them
                for j in list(sid):
                    if new_idx[sid[j]] < 0:
                        del sid[j]
                        dead[j] = True
                    else:
                        sid[j] = int(new_idx[sid[j]])
            gpu_s += sg["solve_seconds"]
            gpu_sweep_s += (t1 - t0) + (t4 - t3)
            rec = {"cameras": ncam, "iterations": sg["iterations"], "rms_px": sg["final_rms_px"], "gpu_solve_ms": 1e3 * sg["solve_seconds"], "gpu_call_ms": 1e3 * (t2 - t1)}
            if er_a or er_b or removed:
                rec["sweep_erased_observations_before_after_removed_landmarks"] = [int(er_a), int(er_b), int(removed)]
            if with_cpu:
                from oracle import orc_ba, orc_validity
                pt_off = np.concatenate([[0], np.cumsum(np.bincount(flat["obs_pt"], minlength=len(flat["points"])))]).astype(np.int32)
                t0 = time.perf_counter()
                i0, k0 = orc_validity.landmark_validity(p34, intr, flat["points"], pt_off, flat["obs_cam"], flat["obs_uv"].astype(np.int32), 1e9, 0.0)
                t1 = time.perf_counter()
                P0, I0, X0, s0 = orc_ba.solve(flat, threads=4)
                t2 = time.perf_counter()
                i1, k1 = orc_validity.landmark_validity(synth_ba.poses_to_34(P0), I0, X0, pt_off, flat["obs_cam"], flat["obs_uv"].astype(np.int32))
                t3 = time.perf_counter()
                cpu_s += s0["solve_seconds"]
                cpu_sweep_s += (t1 - t0) + (t3 - t2)
                ok = (s0["iterations"] == sg["iterations"] and abs(s0["final_rms_px"] - sg["final_rms_px"]) <= 1e-5 and
                      np.array_equal(i0, inl_a) and int((~k0).sum()) == er_a and np.array_equal(i1, inl_b) and int((~k1).sum()) == er_b)
                rec.update({"cpu_solve_ms": 1e3 * s0["solve_seconds"], "equal_to_cpu": bool(ok)})
                equal = equal and ok
            steps.append(rec)
    finally:
        ses.close()
    inc = {"solves": len(steps), "gpu_seconds": gpu_s + gpu_sweep_s, "gpu_solve_seconds": gpu_s, "gpu_sweep_seconds": gpu_sweep_s,
           "lm_iterations": int(sum(r["iterations"] for r in steps)), "per_view": steps}
    if with_cpu:
        inc.update({"cpu_seconds": cpu_s + cpu_sweep_s, "cpu_solve_seconds": cpu_s, "cpu_sweep_seconds": cpu_sweep_s, "cpu_threads": 4, "equal_to_cpu": bool(equal)})
    out["incremental_ba"] = inc
    out["gpu_seconds"] = out["pair_loop_sift128"]["gpu_seconds"] + inc["gpu_seconds"]
    if with_cpu:
        out["cpu_seconds"] = out["pair_loop_sift128"]["cpu_seconds"] + inc["cpu_seconds"]
        out["equal_to_cpu"] = bool(out["pair_loop_sift128"]["equal_to_cpu"] and out["pair_loop_orb32"]["equal_to_cpu"] and inc["equal_to_cpu"])
        out["note"] = "gpu_seconds / cpu_seconds: the SIFT-shaped pair loop + the 23 solves with their sweeps (the ORB-shaped pair loop is listed beside it)"
    return out


class PinnedLists:
    """Pinned host buffers (rcn_host_alloc) for the materialised match lists; two alternate so that the host
    could still read step k's lists while step k+1's copy is in flight."""

    def __init__(self, ctx, n_pairs, capacity):
        import ctypes as C
        self.ctx, self.C = ctx, C
        self.capacity = int(capacity)
        self.offs = [np.zeros(n_pairs + 1, np.int64) for _ in range(2)]
        self.ptr = []
        for _ in range(2):
            p = C.c_void_p()
            ctx.check(ctx.lib.rcn_host_alloc(C.byref(p), 8 * max(1, self.capacity)))
            self.ptr.append(p)
        self.k = 0
        self.total = C.c_int64(0)

    def begin(self, table_ptr, stride, counts_ptr, n_pairs):
        C = self.C
        b = self.k & 1
        self.k += 1
        self.ctx.check(self.ctx.lib.rcn_match_compact_begin(self.ctx.h, C.c_void_p(table_ptr), stride, C.c_void_p(counts_ptr), n_pairs,
                                                            self.offs[b].ctypes.data, self.ptr[b], self.capacity, C.byref(self.total)))
        return self.total.value

    def wait(self):
        self.ctx.check(self.ctx.lib.rcn_match_compact_wait(self.ctx.h))

    def close(self):
        for p in self.ptr:
            self.ctx.lib.rcn_host_free(p)


def parity_cfg2(out, counts):
    """Every pair of the cfg2 table against the CPU oracle's: row hash + count per pair, tests/golden/match_cfg2_full.npz."""
    from reconstructor_amd import tablehash
    path = os.path.join(ROOT, "tests", "golden", "match_cfg2_full.npz")
    if not os.path.exists(path):
        return {"equal_to_cpu": None, "note": "fixture missing"}
    g = np.load(path)
    P, K = len(g["pairs"]), int(g["K"])
    h, c = tablehash.row_hashes(out[:P].cpu().numpy(), K)
    return {"equal_to_cpu": bool(np.array_equal(h, g["hashes"]) and np.array_equal(c, g["counts"]) and np.array_equal(counts[:P].cpu().numpy(), g["counts"])),
            "pairs_compared": int(P), "matches_found_cpu": int(g["matches_found"]),
            "source": "tests/golden/match_cfg2_full.npz (oracle/match_oracle.c over all 4950 pairs: 64-bit row hash + count per pair)"}


def parity_cfg3(world, rank):
    """The sampled pairs of cfg3 this rank computed (pair number p -> rank p % world) against the CPU oracle by row hash + count: 512
    pairs, 64 from every residue of the pair number modulo 8 (tests/golden/match_cfg3_sample512.npz), and 5120 more, 40 for every
    sixteenth of the pair list x residue (match_cfg3_sample5k.npz: the pipeline chunks of the one-GPU call, the ranks of an 8-GPU node)."""
    def check(out, counts):
        import torch
        from reconstructor_amd import tablehash
        ok, n = True, 0
        for name in ("match_cfg3_sample512.npz", "match_cfg3_sample5k.npz"):
            path = os.path.join(ROOT, "tests", "golden", name)
            if not os.path.exists(path):
                return {"equal_to_cpu": None, "note": "fixture missing: " + name}
            g = np.load(path)
            mine = np.nonzero(g["pair_numbers"] % world == rank)[0]
            rows = torch.from_numpy(g["pair_numbers"][mine] // world).to(out.device)          # row of pair p in this rank's table
            h, c = tablehash.row_hashes(out[rows].cpu().numpy(), int(g["K"]))
            ok = ok and bool(np.array_equal(h, g["hashes"][mine]) and np.array_equal(c, g["counts"][mine]) and np.array_equal(counts[rows].cpu().numpy(), g["counts"][mine]))
            n += int(len(mine))
        return {"equal_to_cpu": ok, "pairs_compared": n}
    return check


def gather_leg(torch, dist, dev, shard, out, counts, K, P, n_img, world, rank, fence, reps=2):
    """After the timed steps: the lists of EVERY rank on rank 0 -- the reference's single featureMatches map
    (SequentialReconstructor.cpp:224,264,274) -- through rcn_shard_gather_lists: compaction on every GPU, RCCL send / receive
    device to device, canonical interleave on the root's GPU, one copy into pinned host memory.  Timed on its own (max over
    ranks), bounded by the shard's timeout; a failure is reported in the line, never raised: the step figures stand without it."""
    import ctypes as C
    ctx = shard.ctx
    res = {"root": 0}
    try:
        shard.set_timeout(180.0)
        mine = int(counts[:P].sum().item()) if P else 0
        tot = torch.tensor([float(mine)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tot)
        total = int(tot.item())
        Ptot = n_img * (n_img - 1) // 2
        hp, offs = C.c_void_p(), None
        if rank == 0:
            ctx.check(ctx.lib.rcn_host_alloc(C.byref(hp), 8 * max(1, total)))
            offs = np.zeros(Ptot + 1, np.int64)
        got = C.c_int64(0)
        times = []
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            ctx.check(ctx.lib.rcn_shard_gather_lists(shard.h, 0, C.c_void_p(out.data_ptr()), K, C.c_void_p(counts.data_ptr()),
                                                     offs.ctypes.data if rank == 0 else None, hp if rank == 0 else None, total if rank == 0 else 0, C.byref(got)))
            fence()
            times.append(time.perf_counter() - t0)
        t = torch.tensor([min(times)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res.update({"gather_ms": 1e3 * float(t.item()), "entries": int(got.value), "bytes_to_root_host": 8 * int(got.value)})
        if rank == 0:
            # the root's own pairs sit at numbers 0, world, 2 world, ... of the merged list
            own = counts[:P].cpu().numpy()
            res["consistent"] = bool(got.value == total and offs[-1] == total and np.array_equal(np.diff(offs)[0::world][:P], own))
            ctx.lib.rcn_host_free(hp)
    except Exception as e:      # noqa: BLE001 -- the bench line must still come out
        res["gather_error"] = str(e)[:300]
    return res


def run_grid(torch, dist, dev, shard, n_img, K, local_dev, steps, warmup, world, materialise=True, check=None, gather_rank=None):
    """warmup + `steps` timed steps of exchange -> match -> materialise through the sharded-grid ABI.
    Returns (seconds of the timed region on this rank, stats of the timed steps, matches found, lists bytes).
    gather_rank: this process's rank when the gather leg is to run behind the timed region (st["gather"])."""
    ctx = shard.ctx
    shard.reserve(n_img, K, D)
    info0 = shard.info()
    P = info0["n_pairs"]
    out = torch.empty((max(P, 1), K), dtype=torch.int32, device=dev)
    counts = torch.zeros((max(P, 1),), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    lists = None

    def fence():
        ctx.check(ctx.lib.rcn_synchronize(ctx.h))
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def step():
        nonlocal lists
        shard.exchange(local_dev.data_ptr())
        shard.match(0.7, out.data_ptr(), K, counts.data_ptr())
        if materialise and P > 0:
            if lists is None:          # first step sizes the pinned buffers (same data every step: same total)
                ctx.check(ctx.lib.rcn_synchronize(ctx.h))
                lists = PinnedLists(ctx, P, int(counts[:P].sum().item()))
            lists.begin(out.data_ptr(), K, counts.data_ptr(), P)

    for _ in range(warmup):
        step()
    if lists is None and materialise and P > 0:      # warmup 0: size the buffers outside the timed region
        step()
    if lists is not None:
        lists.wait()
    fence()
    from reconstructor_amd.matcher import HipL2Matcher
    m = HipL2Matcher(ctx=ctx)
    m.stats()                 # clears counters
    m.profile(True)
    shard.profile(True)       # HIP events around exchange / fp32 gather / match, on the streams they run on
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if lists is not None:
        lists.wait()          # the last step's lists have landed in host memory
    fence()
    dt = time.perf_counter() - t0
    st = m.stats()
    m.profile(False)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    st["hbm_used_bytes"] = int(total_b - free_b)          # everything this process holds on the device right after the timed steps
    st["shard_times"] = shard.times()
    shard.profile(False)
    n_matches = int(counts[:P].sum().item()) if P else 0
    list_bytes = 8 * (lists.total.value if lists is not None else 0)
    if lists is not None:
        lists.close()
    st["parity"] = check(out, counts) if check is not None else None      # outside the timed region: the table is still in HBM
    info = shard.info()
    st["gather"] = gather_leg(torch, dist, dev, shard, out, counts, K, P, n_img, world, gather_rank, fence) if gather_rank is not None else None
    return dt, st, n_matches, list_bytes, info


def host_boundary_leg(torch, dev, shard, images, steps=5):
    """The same step entered through the HOST side of the boundary: every image handed over as host rows
    (rcn_shard_put_image: one H2D copy per image, pageable memory as a detector's std::vector would be, then
    pinned), exchange, match, lists back in host memory.  The PCIe-inclusive rate; never `value`."""
    import ctypes as C
    ctx = shard.ctx
    n, K = images.shape[0], images.shape[1]
    P = n * (n - 1) // 2
    out = {}
    pin = C.c_void_p()
    ctx.check(ctx.lib.rcn_host_alloc(C.byref(pin), images.nbytes))
    pinned = np.ctypeslib.as_array((C.c_float * images.size).from_address(pin.value)).reshape(images.shape)
    pinned[...] = images
    try:
        for name, src in (("pageable", images), ("pinned", pinned)):
            shard.reserve(n, K, D)
            lists = None
            def step():
                nonlocal lists
                for i in range(n):
                    shard.put_image(i, src[i])
                shard.exchange(None)
                shard.match(0.7)                      # tables owned by the ctx
                off = np.zeros(P + 1, np.int64)
                if lists is None:
                    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
                    lists = (np.zeros((8 << 20, 2), np.int32), off)      # room for 8 Mi matches (cfg2 finds 2.5 M)
                tot = C.c_int64(0)
                ctx.check(ctx.lib.rcn_shard_lists(shard.h, lists[1].ctypes.data, lists[0].ctypes.data, lists[0].shape[0], C.byref(tot)))
                return tot.value
            step()
            ctx.check(ctx.lib.rcn_synchronize(ctx.h))
            t0 = time.perf_counter()
            for _ in range(steps):
                m = step()
            ctx.check(ctx.lib.rcn_synchronize(ctx.h))
            dt = (time.perf_counter() - t0) / steps
            out[name] = {"ms_per_step": 1e3 * dt, "value": float(P) * K * K / dt, "unit": "pair-distances/s", "matches_found": int(m)}
    finally:
        ctx.lib.rcn_host_free(pin)
    out["note"] = ("host rows in (%.0f MB per step over PCIe, one synchronous copy per image), host lists out; "
                   "the resident-input step is `value` / `cfg2.value`" % (images.nbytes / 1e6))
    return out


def grid_line(K, n_img, n_pairs_total, dt, steps, st, world, my_pairs):
    """value / roofline of one measured grid (pair-distances of the whole job per second; K1 against the f16 MFMA peak)."""
    pd_job = float(n_pairs_total) * K * K
    calls = max(1, st["profiled_calls"])
    launches = max(1, st.get("coarse_launches", 0) or calls)        # pipeline chunks: several launches of K1 per step on large grids
    coarse_ms = st["coarse_ms"] / launches                           # average launch duration (what rocprofv3's kernel stats show)
    flops_per_launch = 2.0 * D * float(st["pair_distances"]) * calls / launches
    flops = flops_per_launch                            # 2*D flop per pair-distance x the pair-distances of one launch (this rank)
    achieved = flops / (coarse_ms * 1e-3) / 1e12 if coarse_ms > 0 else 0.0
    traffic, tnote = measured_traffic("k_coarse_top2<256>@%dx%d" % (n_img, K)) if world == 1 else (None, "PMC passes are single-GPU")
    roof = {"bound": "mfma", "kernel": "k_coarse_top2<256, 0, 1> (v_mfma_f32_16x16x32_f16)", "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": traffic,
            "traffic_note": "HBM bytes per launch from rocprofv3 --pmc passes on this exact source (profiles/r05_match_traffic.json); " + str(tnote),
            # per LAUNCH like `traffic`: every resident fp16 image once (a pipeline chunk's pairs meet nearly all of them as train images)
            # + 8 B of candidate pair per query row of the launch's pairs
            "algorithmic_hbm_bytes_per_launch": float(n_img) * K * D * 2 + 8.0 * float(st["rows_total"]) * calls / launches,
            "launch_ms": coarse_ms, "launches_per_step": launches / calls, "k1_ms_per_step": st["coarse_ms"] / calls,
            # K1's time per image pair of THIS rank: the same at every N unless something competes with the kernel (at N > 1 the fp32 rows
            # arrive on a side stream beside it: any interference of that gather shows here against the N = 1 line's figure)
            "k1_us_per_pair_rank0": 1e3 * st["coarse_ms"] / calls / max(1, my_pairs),
            "algorithmic_flop_per_launch": flops_per_launch,
            "rerank_ms": st["rerank_ms"] / calls, "unique_ms": st["unique_ms"] / calls,
            "rows_brute_force": int(st.get("rows_brute_force", 0)),
            "source_hash": source_hash()}
    return pd_job * steps / dt, dt / steps * 1e3, roof


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: run the N ranks as a child torch.distributed.run and
    exit with its code.  Nothing here initialises the GPU (device_count() does not), so starting children is safe."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d device(s) visible: refusing to run a smaller world\n" % (args.gpus, have))
        raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def chunk_ab_leg(torch, ctx, dev, n_img, K, n_pairs=32768, reps=3):
    """what the pipeline chunks cost, A/B on THIS box.  The first 32 768 pairs of the resident cfg-3 grid (1.34e8 query rows:
    exactly the default workspace) as one launch of K1, as two and as four (rcn_match_set_workspace_rows); tables left in HBM."""
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    m = HipL2Matcher(ctx=ctx)
    pairs = all_pairs(n_img)[:n_pairs]
    out = torch.empty((len(pairs), K), dtype=torch.int32, device=dev)
    cnt = torch.empty((len(pairs),), dtype=torch.int32, device=dev)
    res = []
    try:
        for rows in (1 << 27, 1 << 26, 1 << 25):
            m.set_workspace_rows(rows)
            m.match_grid_device(pairs, out.data_ptr(), K, cnt.data_ptr())
            ctx.check(ctx.lib.rcn_synchronize(ctx.h))
            m.stats()
            m.profile(True)
            t0 = time.perf_counter()
            for _ in range(reps):
                m.match_grid_device(pairs, out.data_ptr(), K, cnt.data_ptr())
            ctx.check(ctx.lib.rcn_synchronize(ctx.h))
            dt = (time.perf_counter() - t0) / reps
            st = m.stats()
            m.profile(False)
            res.append({"chunks": int(st["chunks"]), "k1_ms": st["coarse_ms"] / max(1, st["profiled_calls"]), "call_ms": 1e3 * dt})
    finally:
        m.set_workspace_rows(0)
        m.profile(False)
    one = res[0]["k1_ms"]
    return {"pairs": int(len(pairs)), "runs": res, "k1_single_launch_ms": one,
            "k1_cost_of_chunks": {str(r["chunks"]): r["k1_ms"] / one - 1.0 for r in res[1:]},
            "note": "same box, same data: K1's time for the first %d pairs of the grid as 1 / 2 / 4 launches; the full grid's sixteen chunks are this size" % len(pairs)}


def small_d_leg(torch, ctx, dev, kind, Dd, n3=100, K3=1500):
    """K1 on the descriptor lengths the reference's classic detectors produce: a cfg2-sized grid (100 images x 1500 keypoints,
    4950 pairs) of SIFT-like 128-d or ORB-as-float 32-d rows, tables left in HBM; roofline of the coarse kernel against the
    f16 MFMA peak with 2 D flop per pair-distance."""
    from reconstructor_amd import synth
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    pool3 = synth.world_pool(kind, 4 * K3, seed=1234)
    loc3 = np.stack([synth.image_descriptors(kind, i, K3, pool3, seed=1234) for i in range(n3)])
    loc3_dev = torch.from_numpy(loc3).to(dev)
    m3 = HipL2Matcher(ctx=ctx)
    m3.clear()
    m3.upload_batch_device(0, n3, loc3_dev.data_ptr(), K3, Dd)
    pr3 = all_pairs(n3)
    o3 = torch.empty((len(pr3), K3), dtype=torch.int32, device=dev)
    c3 = torch.empty((len(pr3),), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    for _ in range(3):
        m3.match_grid_device(pr3, o3.data_ptr(), K3, c3.data_ptr())
    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    m3.stats(); m3.profile(True)
    t3 = time.perf_counter()
    for _ in range(20):
        m3.match_grid_device(pr3, o3.data_ptr(), K3, c3.data_ptr())
    ctx.check(ctx.lib.rcn_synchronize(ctx.h))
    dt3 = (time.perf_counter() - t3) / 20
    st3 = m3.stats(); m3.profile(False)
    cms3 = st3["coarse_ms"] / max(1, st3["profiled_calls"])
    pd3 = float(st3["pair_distances"])
    ach = 2.0 * Dd * pd3 / (cms3 * 1e-3) / 1e12
    out = {"workload": "100 images x 1500 %s keypoints x %d-d, 4950 image pairs (tables left in HBM)" % ("SIFT-like" if kind == "sift" else "ORB-as-float", Dd),
           "value": pd3 / dt3, "unit": "pair-distances/s", "ms_per_step": 1e3 * dt3, "matches_found": int(c3.sum().item()),
           "rows_reranked": int(st3["rows_reranked"]), "rows_exact_fallback": int(st3["rows_exact_fallback"]), "rows_total": int(st3["rows_total"]),
           "roofline": {"bound": "mfma", "kernel": "k_coarse_top2<%d, 0, 0> (v_mfma_f32_32x32x16_f16, %d train rows per stage)" % (Dd, 128), "achieved": ach,
                        "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F16_PEAK_TFLOPS,
                        "launch_ms": cms3, "traffic": measured_traffic("k_coarse_top2<%d>@100x1500" % Dd)[0],
                        "note": "2 D flop per pair-distance: at small D the top-2 fold (3 vector operations per pair-distance, whatever D) outweighs the MFMAs, so the fraction of the MFMA peak falls with D by construction"}}
    m3.clear()
    del loc3_dev, o3, c3
    return out


def mixed_magnitude_leg(torch, ctx, dev, n3=100, K3=1500):
    """What ONE out-of-scale image costs (the fp16 scale is one power of two for all resident images).  The sift128 grid again,
    with every row of image 50 multiplied by 2^30.  Round 2 let the scale follow the largest row: the other 99 images' fp16
    copies would underflow, nothing would certify and all 7.4 M rows would go through the exact kernels.  Now the histogram of
    row norms (fix_scale, match.hip) sets that image's rows aside as BIG rows -- exact kernel as queries, never coarse
    candidates, a norm bound in everybody else's certificates -- and the scale stays where the other 99 images need it.
    Results stay exact (the pairs that do not involve image 50 must equal the unscaled run); reported: throughput and the share
    of rows that went to the exact kernel (expected: image 50's own query rows, 49 x 1500 of 7.4 M)."""
    from reconstructor_amd import synth
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    pool3 = synth.world_pool("sift", 4 * K3, seed=1234)
    loc3 = np.stack([synth.image_descriptors("sift", i, K3, pool3, seed=1234) for i in range(n3)])
    pr3 = all_pairs(n3)
    m3 = HipL2Matcher(ctx=ctx)
    res = {}
    for name, factor in (("uniform", 1.0), ("image_50_times_2p30", 2.0 ** 30)):
        x = loc3.copy()
        x[50] *= np.float32(factor)
        xd = torch.from_numpy(x).to(dev)
        m3.clear()
        m3.upload_batch_device(0, n3, xd.data_ptr(), K3, 128)
        o3 = torch.empty((len(pr3), K3), dtype=torch.int32, device=dev)
        c3 = torch.empty((len(pr3),), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        m3.match_grid_device(pr3, o3.data_ptr(), K3, c3.data_ptr())
        ctx.check(ctx.lib.rcn_synchronize(ctx.h))
        t3 = time.perf_counter()
        for _ in range(3):
            m3.match_grid_device(pr3, o3.data_ptr(), K3, c3.data_ptr())
        ctx.check(ctx.lib.rcn_synchronize(ctx.h))
        dt3 = (time.perf_counter() - t3) / 3
        st3 = m3.stats()
        res[name] = {"value": float(st3["pair_distances"]) / dt3, "ms_per_step": 1e3 * dt3, "rows_total": int(st3["rows_total"]),
                     "rows_reranked": int(st3["rows_reranked"]), "rows_exact_fallback": int(st3["rows_exact_fallback"]), "rows_brute_force": int(st3["rows_brute_force"]),
                     "fallback_share": float(st3["rows_exact_fallback"]) / max(1, float(st3["rows_total"])), "table": o3.cpu().numpy()}
        m3.clear()
        del xd, o3, c3
    keep = (pr3 != 50).all(1)
    same = bool(np.array_equal(res["uniform"]["table"][keep], res["image_50_times_2p30"]["table"][keep]))
    for r in res.values():
        del r["table"]
    return {"workload": "sift128 grid (100 x 1500 x 128, 4950 pairs); second run with image 50 scaled by 2^30", "unit": "pair-distances/s",
            "uniform": res["uniform"], "image_50_times_2p30": res["image_50_times_2p30"], "pairs_without_image_50_identical": same,
            "slowdown": res["uniform"]["value"] / res["image_50_times_2p30"]["value"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--images", type=int, default=0, help="override image count (debug)")
    ap.add_argument("--kpts", type=int, default=0, help="override keypoints per image (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true", help="skip the BA / epipolar / sweep legs")
    ap.add_argument("--no-cfg2", action="store_true", help="skip the configs[1] sub-measurement of the N = 1 line")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: the line would not describe the run\n" % (args.gpus, world))
        raise SystemExit(2)
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d has no device %d (%d visible)\n" % (rank, local_rank, torch.cuda.device_count()))
        raise SystemExit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo")        # control plane only; the data plane is RCCL inside librcn.so

    from reconstructor_amd import _lib, pairgrid, synth

    n_img, K = WORKLOADS[args.workload]
    n_img, K = args.images or n_img, args.kpts or K
    ctx = _lib.Context(local_rank)
    # rendezvous: rank 0 draws the RCCL id, the control plane hands it round
    uid = [pairgrid.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    # RCCL prints its version banner to STDOUT when the first communicator comes up; this process's stdout carries ONE
    # JSON line: send fd 1 to stderr meanwhile
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        shard = pairgrid.Shard(ctx, rank, world, uid[0])
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)

    # ---- inputs: every rank generates only the images it "detected"
    lo, hi = pairgrid.owned_images(n_img, world, rank)
    pool = synth.world_pool("superpoint", 4 * K, seed=1234)
    local = np.stack([synth.image_descriptors("superpoint", i, K, pool, seed=1234) for i in range(lo, hi)]) if hi > lo \
        else np.zeros((0, K, D), np.float32)
    local_dev = torch.from_numpy(local).to(dev) if hi > lo else torch.zeros((1, K, D), dtype=torch.float32, device=dev)
    n_pairs_total = n_img * (n_img - 1) // 2

    is_cfg = (n_img, K) == WORKLOADS[args.workload]
    main_check = parity_cfg3(world, rank) if (is_cfg and args.workload == "cfg3") else (parity_cfg2 if (is_cfg and args.workload == "cfg2" and world == 1) else None)
    dt, st, n_matches, list_bytes, info = run_grid(torch, dist, dev, shard, n_img, K, local_dev, args.steps, args.warmup, world, check=main_check, gather_rank=rank)
    par = st.get("parity") or {"equal_to_cpu": None, "pairs_compared": 0}

    tm = st["shard_times"]
    phase = [tm["exchange_ms"] / max(1, tm["exchanges"]), tm["f32_gather_ms"] / max(1, tm["exchanges"]), tm["match_ms"] / max(1, tm["matches"])]
    if world > 1:
        t = torch.tensor([dt] + phase, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        phase = [float(x) for x in t[1:].tolist()]
        tot = torch.tensor([float(n_matches), float(list_bytes), float(st["rows_reranked"]), float(st["rows_exact_fallback"]), float(st["rows_total"]),
                            float(par["pairs_compared"]), 0.0 if par["equal_to_cpu"] in (True, None) else 1.0], dtype=torch.float64)
        dist.all_reduce(tot)
        n_matches, list_bytes = int(tot[0].item()), int(tot[1].item())
        rows = [int(x) for x in tot[2:5].tolist()]
        if par["equal_to_cpu"] is not None:
            par = {"equal_to_cpu": tot[6].item() == 0.0, "pairs_compared": int(tot[5].item())}
    else:
        rows = [int(st["rows_reranked"]), int(st["rows_exact_fallback"]), int(st["rows_total"])]

    if rank == 0:
        value, ms_step, roof = grid_line(K, n_img, n_pairs_total, dt, args.steps, st, world, info["n_pairs"])
        name = args.workload if (n_img, K) == WORKLOADS[args.workload] else "custom"
        line = {
            "metric": "descriptor pair-distances/s + BA LM-iterations/s (1k cams, 100k pts)",
            "metric_note": "value = 256-d L2 pair-distances/s of the whole step: descriptor exchange (RCCL) + exact 2-NN + ratio + uniqueness over all image pairs + (query, train) lists materialised in host memory; BA LM-iterations/s are in `ba` (single GPU: BA does not shard)",
            "value": value, "unit": "pair-distances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16+f64",
            "dtype_note": "f16 MFMA (f32 accumulate) coarse pass, f64 exact re-rank; indices bit-exact vs the f64 oracle",
            "data": "synthetic",
            "config": {"workload": "%s: %d images x %d keypoints x %d-d, %d image pairs, the same grid at every N (pair number p -> rank p %% N)"
                                   % (name, n_img, K, D, n_pairs_total),
                       "pairs_per_rank": int(info["n_pairs"]), "pair_matches_per_s": n_pairs_total * args.steps / dt,
                       "matches_found": n_matches, "host_list_bytes_per_step": list_bytes,
                       "exchange_bytes_f16_payload": int(info["exchange_bytes_f16"]), "exchange_bytes_f32_side_stream": int(info["exchange_bytes_f32"]),
                       "rows_reranked": rows[0], "rows_exact_fallback": rows[1], "rows_total": rows[2],
                       "hbm_used_gb_rank0": st.get("hbm_used_bytes", 0) / 1e9, "pipeline_chunks": int(st.get("chunks", 1)),
                       # the table of the LAST timed step against the CPU oracle (outside the timed region): cfg3 = 5632 sampled
                       # pairs by row hash + count (every residue of the pair number mod 8 x every sixteenth of the pair list), over all ranks; cfg2 = all 4950 pairs
                       "equal_to_cpu": par["equal_to_cpu"], "pairs_compared_with_cpu": par["pairs_compared"]},
            "roofline": roof,
            # the collective side of the step, per step, max over ranks (HIP events inside librcn.so): `ranks` is
            # ncclCommCount of the communicator the gathers ran on; the fp32 gather runs on a side stream BESIDE the
            # coarse kernel (it is inside match_ms, not added to it)
            "rccl": {"ranks": int(info["comm_ranks"]), "exchange_ms": phase[0], "f32_gather_ms_side_stream": phase[1], "match_ms": phase[2],
                     "host_syncs_per_exchange": 1},
        }
        # SURVEY 8(e) "per-pair match lists gathered to rank 0": measured behind the timed steps (rcn_shard_gather_lists, max over
        # ranks); `value` keeps every rank's lists on that rank, value_with_gather adds the gather to every step
        g = st.get("gather") or {}
        line["rccl"].update({k: g[k] for k in ("gather_ms", "entries", "bytes_to_root_host", "consistent", "gather_error", "root") if k in g})
        if "gather_ms" in g:
            line["rccl"]["value_with_gather"] = float(n_pairs_total) * K * K / (1e-3 * (ms_step + g["gather_ms"]))
            line["rccl"]["gather_note"] = ("all ranks' (query, train) lists in canonical pair order in rank 0's pinned host memory: compaction on every GPU, ncclSend / ncclRecv "
                                           "device to device, interleave on the root's GPU, one D2H copy; `value` is quoted without it (each rank's lists in its own host memory)")
        if world == 1:
            if is_cfg and args.workload == "cfg3":
                line["roofline"]["chunk_ab"] = chunk_ab_leg(torch, ctx, dev, n_img, K)
            if not args.no_cfg2 and args.workload != "cfg2":
                # BASELINE configs[1] on the same GPU, same path (exchange + match + materialise)
                n2, K2 = WORKLOADS["cfg2"]
                pool2 = synth.world_pool("superpoint", 4 * K2, seed=1234)
                loc2 = np.stack([synth.image_descriptors("superpoint", i, K2, pool2, seed=1234) for i in range(n2)])
                loc2_dev = torch.from_numpy(loc2).to(dev)
                # three windows of 20 steps, the median one is reported (a window is 0.17 s: one clock transient moves it by 10 %)
                runs2 = sorted((run_grid(torch, dist, dev, shard, n2, K2, loc2_dev, 20, 3, 1, check=parity_cfg2) for _ in range(3)), key=lambda r: r[0])
                dt2, st2, nm2, lb2, info2 = runs2[1]
                v2, ms2, roof2 = grid_line(K2, n2, n2 * (n2 - 1) // 2, dt2, 20, st2, 1, info2["n_pairs"])
                # and with the tables left in HBM (round 1's definition of the step), for comparison
                dt2b, st2b, _, _, _ = run_grid(torch, dist, dev, shard, n2, K2, loc2_dev, 20, 3, 1, materialise=False)
                line["cfg2"] = {"workload": "cfg2: %d images x %d keypoints x %d-d, %d image pairs" % (n2, K2, D, n2 * (n2 - 1) // 2),
                                "value": v2, "unit": "pair-distances/s", "steps": 20, "warmup": 3, "windows": "median of 3", "ms_per_step": ms2, "matches_found": nm2,
                                "equal_to_cpu": st2["parity"]["equal_to_cpu"], "parity": st2["parity"],
                                "host_list_bytes_per_step": lb2, "roofline": roof2,
                                "value_tables_left_in_hbm": float(n2 * (n2 - 1) // 2) * K2 * K2 * 20 / dt2b,
                                "rows_reranked": int(st2["rows_reranked"]), "rows_exact_fallback": int(st2["rows_exact_fallback"]), "rows_total": int(st2["rows_total"])}
                line["cfg2"]["host_boundary"] = host_boundary_leg(torch, dev, shard, loc2)
                del loc2_dev
                # the reference's ACTIVE descriptor is SIFT (128-d, FeatureDetector.cpp:9-10; its README's 76 s matching stage is
                # 100 images), the commented-out one ORB, matched as 32 floats (:19-24): K1 at D = 128 and D = 32 on cfg2-sized grids
                line["sift128"] = small_d_leg(torch, ctx, dev, "sift", 128)
                line["orb32"] = small_d_leg(torch, ctx, dev, "orb", 32)
                line["mixed_magnitudes"] = mixed_magnitude_leg(torch, ctx, dev)
                shard.reserve(n_img, K, D)         # the clear above dropped the shard's images: nothing is matched through it again
            if not args.no_ba:
                shard.close()
                shard = None
                line["ba"] = ba_leg(ctx, not args.no_cpu_baseline)
                line["ba_lm_iterations_per_s"] = line["ba"]["cfg5"]["lm_iterations_per_s"]   # 1k cams / 100k pts / 1M obs
                line["epipolar_filter"] = fmat_leg(ctx, not args.no_cpu_baseline)
                line["cfg1"] = cfg1_leg(ctx, not args.no_cpu_baseline)
        # the CPU baseline rides in EVERY line: rank 0 times the oracle on the same bounded sample of its own images behind
        # the timed region while the other ranks wait at the closing barrier (at N > 1 they are idle: the host cores are rank 0's)
        if not args.no_cpu_baseline:
            images = [local[i] for i in range(min(local.shape[0], 64))]
            line["cpu_baseline"] = cpu_baseline(images, host_threads())
        else:
            line["cpu_baseline"] = None
        if info["comm_ranks"] != world:
            sys.stderr.write("bench.py: the communicator has %d ranks, the job %d\n" % (info["comm_ranks"], world))
            raise SystemExit(3)
        line["config"]["rccl_ranks"] = int(info["comm_ranks"])
        print(json.dumps(line), flush=True)
    if shard is not None:
        shard.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

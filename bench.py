#!/usr/bin/env python3
"""bench.py -- headline benchmark: descriptor pair-distances/s of the all-pairs matcher.

Workload (BASELINE.json configs[1]): synthetic 100 images x 2048 SuperPoint-like 256-d
keypoints, exact brute-force L2 2-NN + ratio + uniqueness over all 4950 image pairs.
One "step" = one pass of the hot path over the whole pair grid, descriptors resident in HBM,
match tables left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 (weak scaling): the image count grows as 100*sqrt(N) so every rank keeps ~4950 pairs.
Each rank owns n/N images ("detected locally"), one RCCL all-gather over xGMI replicates
the descriptors, then the pair list is dealt round-robin to ranks; no other exchange.  Every
timed step gathers, ingests and matches one whole batch; the all-gather of batch k+1 runs on a
side stream into a second landing buffer while batch k is matched (double buffering).
Prints ONE JSON line on rank 0.
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

K_PER_IMAGE = 2048
D = 256
N_IMAGES_1GPU = 100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense bf16/f16 MFMA peak (same guide)


def host_threads():
    """Threads for the CPU baselines: the GPU box's CPU share for one GPU is 16 cores even though
    the affinity mask lists every core of the host."""
    return max(1, min(16, len(os.sched_getaffinity(0))))


def cpu_baseline(images, n_threads):
    """Oracle (CPU restatement, kind "port") on a bounded sample of the same workload: about ten
    seconds at all of the box's cores for this GPU, plus the reference's own thread count
    (MAX_NUM_THREADS = 4, SequentialReconstructor.h:17) on a smaller sample."""
    from oracle import orc
    allp = orc.all_pairs(len(images))
    rng = np.random.default_rng(0)

    def timed(n_pairs, threads):
        pick = allp[rng.choice(len(allp), size=min(n_pairs, len(allp)), replace=False)]
        t0 = time.perf_counter()
        orc.match_grid(images, pick, threads=threads)
        dt = time.perf_counter() - t0
        return sum(images[a].shape[0] * images[b].shape[0] for a, b in pick) / dt, len(pick), dt

    orc.match_grid(images, allp[:n_threads], threads=n_threads)  # warm (page-in, transposes)
    v, n, dt = timed(100 * n_threads, n_threads)
    v4, n4, dt4 = timed(96, 4)
    return {"value": v, "unit": "pair-distances/s", "cores": n_threads, "kind": "port",
            "sample": "%d of %d image pairs (2048x2048x256 each), oracle/match_oracle.c, OpenMP over pairs, %.1f s"
                      % (n, len(allp), dt),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=0, help="override image count (debug)")
    ap.add_argument("--kpts", type=int, default=0, help="override keypoints per image (debug, e.g. 4096 = cfg3 shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true", help="skip the BA leg (cfg 4 + cfg 5 LM iterations/s)")
    args = ap.parse_args()

    global K_PER_IMAGE
    if args.kpts:
        K_PER_IMAGE = args.kpts
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    # REHEARSAL ONLY (one-GPU box): RCN_BENCH_REHEARSE=1 maps every rank to device 0 and moves the
    # all-gather through the host with gloo, so the rest of the N>1 path can be exercised.
    rehearse = os.environ.get("RCN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from reconstructor_amd import synth
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    from reconstructor_amd import pairgrid

    n_img = args.images or int(round(N_IMAGES_1GPU * math.sqrt(world)))
    n_img = (n_img + world - 1) // world * world if world > 1 else n_img
    matcher = HipL2Matcher(device=local_rank)
    stream = torch.cuda.current_stream(dev)
    matcher.ctx.check(matcher.ctx.lib.rcn_set_stream(matcher.ctx.h, stream.cuda_stream))

    # ---- inputs: every rank generates only the images it "detected"
    lo, hi = pairgrid.owned_images(n_img, world, rank)
    pool = synth.world_pool("superpoint", 4 * K_PER_IMAGE, seed=1234)
    local = np.stack([synth.image_descriptors("superpoint", i, K_PER_IMAGE, pool, seed=1234)
                      for i in range(lo, hi)])
    local_dev = torch.from_numpy(local).to(dev)
    pairs = all_pairs(n_img)
    my_pairs = pairgrid.shard_pairs(pairs, world, rank)
    out = torch.empty((len(my_pairs), K_PER_IMAGE), dtype=torch.int32, device=dev)
    counts = torch.empty((len(my_pairs),), dtype=torch.int32, device=dev)
    # N > 1: two landing buffers; the all-gather of batch k+1 runs on a side stream (RCCL over
    # xGMI) while batch k is being matched -- every step still gathers, ingests and matches
    # one whole batch inside the timed region.
    gathered = [torch.empty((n_img, K_PER_IMAGE, D), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    pending = [None, None]
    state = {"k": 0, "overlap": world > 1 and not rehearse}

    def launch_gather(slot):
        """Start the all-gather of the next batch into gathered[slot] (asynchronous)."""
        if rehearse:
            host = torch.empty(gathered[slot].shape, dtype=torch.float32)
            dist.all_gather_into_tensor(host.view(-1), local_dev.cpu().view(-1))
            gathered[slot].copy_(host)
            pending[slot] = None
            return
        if state["overlap"]:
            try:
                comm_stream.wait_stream(torch.cuda.current_stream(dev))   # slot's previous readers are queued before this point
                with torch.cuda.stream(comm_stream):
                    pending[slot] = dist.all_gather_into_tensor(gathered[slot].view(-1), local_dev.view(-1), async_op=True)
                return
            except Exception as exc:   # fall back to the in-line collective
                state["overlap"] = False
                sys.stderr.write("bench: async all-gather unavailable (%s); using the in-line collective\n" % exc)
        dist.all_gather_into_tensor(gathered[slot].view(-1), local_dev.view(-1))
        pending[slot] = None

    if world == 1:
        matcher.upload_batch_device(0, n_img, local_dev.data_ptr(), K_PER_IMAGE, D)
    else:
        launch_gather(0)

    def step():
        if world > 1:
            slot = state["k"] & 1
            if pending[slot] is not None:
                pending[slot].wait()                      # main stream waits for the collective
                pending[slot] = None
            matcher.upload_batch_device(0, n_img, gathered[slot].data_ptr(), K_PER_IMAGE, D)
            launch_gather(slot ^ 1)                       # next batch travels while this one is matched
            state["k"] += 1
        matcher.match_grid_device(my_pairs, out.data_ptr(), K_PER_IMAGE, counts.data_ptr())

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    matcher.stats()                 # clears counters
    matcher.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    st = matcher.stats()
    matcher.profile(False)

    if world > 1:
        rdev = torch.device("cpu") if rehearse else dev
        tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([float(st["pair_distances"]), float(counts.sum().item())], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot)
        pd_step, n_matches = float(tot[0].item()), int(tot[1].item())
    else:
        pd_step, n_matches = float(st["pair_distances"]), int(counts.sum().item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = pd_step * args.steps / dt
        calls = max(1, st["profiled_calls"])
        coarse_ms = st["coarse_ms"] / calls
        my_pd = float(st["pair_distances"])
        flops = 2.0 * D * my_pd                      # 2*D flop per pair-distance
        achieved = flops / (coarse_ms * 1e-3) / 1e12 if coarse_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_match_traffic.json")
        if world == 1 and n_img == N_IMAGES_1GPU and os.path.exists(tpath):
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]   # PMC passes, see the file
        line = {
            "metric": "descriptor pair-distances/s + BA LM-iterations/s (1k cams, 100k pts)",
            "metric_note": "value = 256-d L2 pair-distances/s of the whole step (exact 2-NN + ratio + uniqueness over all image pairs); BA LM-iterations/s are in `ba` (single GPU: BA does not shard)",
            "value": value, "unit": "pair-distances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16+f64",
            "dtype_note": "f16 MFMA (f32 accumulate) coarse pass, f64 exact re-rank; indices bit-exact vs the f64 oracle",
            "data": "synthetic",
            "config": {"workload": ("cfg2" if K_PER_IMAGE == 2048 else "custom") + ": %d images x %d keypoints x %d-d, %d image pairs%s"
                                   % (n_img, K_PER_IMAGE, D, len(pairs),
                                      "" if world == 1 else " (weak scaling of cfg2: ~4950 pairs per rank, RCCL all-gather (double-buffered, overlapped with the previous batch's matching) + ingest inside the step)"),
                       "pairs_per_rank": int(len(my_pairs)), "pair_matches_per_s": len(pairs) * args.steps / dt,
                       "matches_found": n_matches,
                       "rows_reranked": int(st["rows_reranked"]), "rows_exact_fallback": int(st["rows_exact_fallback"]), "rows_total": int(st["rows_total"])},
            "roofline": {"bound": "mfma", "kernel": "k_coarse_top2<256>", "achieved": achieved,
                         "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "bytes per launch from rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes (profiles/r01_match_traffic.json); kernel is MFMA-bound, algorithmic HBM bytes per launch = 1.86e8",
                         "launch_ms": coarse_ms, "rerank_ms": st["rerank_ms"] / calls,
                         "unique_ms": st["unique_ms"] / calls},
        }
        if not args.no_cpu_baseline and world == 1:
            images = [local[i] for i in range(local.shape[0])]
            line["cpu_baseline"] = cpu_baseline(images, host_threads())
        elif world == 1:
            line["cpu_baseline"] = None
        if not args.no_ba and world == 1:
            line["ba"] = ba_leg(matcher.ctx, not args.no_cpu_baseline)
            line["ba_lm_iterations_per_s"] = line["ba"]["cfg5"]["lm_iterations_per_s"]   # 1k cams / 100k pts / 1M obs
            line["epipolar_filter"] = fmat_leg(matcher.ctx, not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

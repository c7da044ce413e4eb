"""tools/pmc_report.py <prof dir> <out dir> [tag] -- condense the raw output of tools/profile_round.sh into the files under profiles/
(named <tag>_*, tag = r04 by default):
per-kernel counter means, the derived figures (effective clock, MFMA-pipe busy share, VALU per MFMA, fabric bytes per
launch corrected as MI355X_MICROARCH.md prescribes) and the traffic JSON bench.py quotes.  Everything is stamped with
the hash of the sources the profiled binary was built from (bench.source_hash) and the git HEAD of the build tree."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counters(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def kernel_stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
    return out


def main():
    prof, outd = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r04"
    import bench
    stamp = {"source_hash": bench.source_hash()}
    if os.environ.get("RCN_GIT_HEAD"):          # .git does not travel to the GPU box: the caller hands the head over (gpurun -- 'RCN_GIT_HEAD=... bash tools/profile_round.sh r05')
        stamp["git_head"] = os.environ["RCN_GIT_HEAD"]
        stamp["git_dirty"] = os.environ.get("RCN_GIT_DIRTY", "") not in ("", "0")
    else:
      try:
        stamp["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
        stamp["git_dirty"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "reconstructor_amd", "include"]).decode().strip())
      except Exception:
        pass
    traffic = dict(stamp)
    traffic["_comment"] = ("fabric-side bytes per launch of K1 from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh): counters in KiB; "
                           "on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact; Infinity-Cache hits are "
                           "counted by these counters, so this is an upper bound on DRAM bytes (MI355X_MICROARCH.md, HBM section)")
    lines = ["# condensed by tools/pmc_report.py from gpurun_out/prof_" + tag + " (tools/profile_round.sh); source_hash %s git %s%s" %
             (stamp["source_hash"], stamp.get("git_head", "?")[:12], " +uncommitted" if stamp.get("git_dirty") else "")]
    for shape, label in (("100_2048_256_6", "100x2048"), ("100_1500_128_6", "sift 100x1500x128"), ("1000_4096_256", "1000x4096")):
        sq = counters(os.path.join(prof, "k1_sq_" + shape))
        lds = counters(os.path.join(prof, "k1_lds_" + shape)) if os.path.isdir(os.path.join(prof, "k1_lds_" + shape)) else {}
        fe = counters(os.path.join(prof, "k1_fetch_" + shape))
        wr = counters(os.path.join(prof, "k1_write_" + shape))
        kn = [k for (k, c) in sq if "coarse" in k]
        if not kn:
            continue
        k = kn[0]
        g = lambda tab, c: tab.get((k, c), (float("nan"), 0))[0]
        cyc = g(sq, "GRBM_GUI_ACTIVE") / 8.0
        rec = {"kernel": k.replace("void ", ""), "shape": label, "launches_averaged": sq[(k, "GRBM_GUI_ACTIVE")][1],
               "cycles_per_launch": cyc, "mfma_busy_share": g(sq, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc),
               "valu_per_mfma": g(sq, "SQ_INSTS_VALU") / g(sq, "SQ_INSTS_MFMA"), "insts_mfma": g(sq, "SQ_INSTS_MFMA"),
               "wave_cycles": g(sq, "SQ_WAVE_CYCLES"), "wait_any_share": g(sq, "SQ_WAIT_ANY") / g(sq, "SQ_WAVE_CYCLES") if (k, "SQ_WAIT_ANY") in sq else None,
               "wait_inst_share": g(sq, "SQ_WAIT_INST_ANY") / g(sq, "SQ_WAVE_CYCLES") if (k, "SQ_WAIT_INST_ANY") in sq else None,
               "lds_bank_conflict_cycles": g(lds, "SQ_LDS_BANK_CONFLICT") if lds else None, "lds_idx_active": g(lds, "SQ_LDS_IDX_ACTIVE") if lds else None,
               "fetch_size_kib": g(fe, "FETCH_SIZE"), "write_size_kib": g(wr, "WRITE_SIZE"),
               "traffic_bytes_per_launch": 1024.0 * (2.0 * g(fe, "FETCH_SIZE") + g(wr, "WRITE_SIZE")),
               "note": "FETCH_SIZE x 2 + WRITE_SIZE (KiB), fabric side, Infinity-Cache hits included"}
        # kernel time from the same passes' logs (HIP events inside tools/k1_run.py)
        for lg in glob.glob(os.path.join(prof, "k1_sq_%s.log" % shape)) + glob.glob(os.path.join(prof, "k1_sq_cfg3.log")):
            for ln in open(lg):
                if ln.startswith("K1 ") and ((shape.startswith("1000") and "cfg3" in lg) or (not shape.startswith("1000") and "cfg3" not in lg)):
                    rec["k1_run_line"] = ln.strip()
                    ms = float(ln.split()[1])
                    rec["effective_clock_ghz_profiled"] = cyc / (ms * 1e-3) / 1e9
        key = "k_coarse_top2<%s>@%s" % ("128" if "128" in shape.split("_")[2:3] else "256", label if "x" in label and "sift" not in label else "100x1500")
        traffic[key] = rec
        lines.append("")
        lines.append("## %s  (%s)" % (rec["kernel"], label))
        for kk, vv in rec.items():
            lines.append("%-34s %s" % (kk, vv))
        lines.append("-- raw counter means (per launch)")
        for tab in (sq, lds, fe, wr):
            for (kn2, c), (v, n) in sorted(tab.items()):
                lines.append("%-44s %-28s n=%d mean=%.5g" % (kn2[:44], c, n, v))
    # the fold ablation at the SIFT shape (diagnostic build, RCN_COARSE_ABL): what a cheaper top-2 fold could buy at most
    abl_names = {0: "shipping fold: and_or + med3 + min per element", 2: "values-only fold: med3 + min (results wrong by construction)", 6: "running minimum only: one operation (results wrong by construction)", 1: "no fold at all"}
    if any(os.path.isdir(os.path.join(prof, "fold_abl%d" % a)) for a in abl_names):
        lines.append("")
        lines.append("## top-2 fold ablation, k_coarse_top2<128> at 100 x 1500 x 128 (tools/profile_round.sh, diagnostic build)")
        for a, nm in abl_names.items():
            sqa = counters(os.path.join(prof, "fold_abl%d" % a))
            ks = [k for (k, c) in sqa if "coarse" in k]
            if not ks:
                continue
            ga = lambda c: sqa.get((ks[0], c), (float("nan"), 0))[0]
            cyc = ga("GRBM_GUI_ACTIVE") / 8.0
            k1 = [ln.strip() for ln in open(os.path.join(prof, "fold_abl%d.log" % a)) if ln.startswith("K1 ")] if os.path.exists(os.path.join(prof, "fold_abl%d.log" % a)) else []
            lines.append("RCN_COARSE_ABL=%d  %-62s cycles %.4g  mfma_busy %.3f  valu_per_mfma %.2f  | %s" %
                         (a, nm, cyc, ga("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), ga("SQ_INSTS_VALU") / ga("SQ_INSTS_MFMA"), k1[-1] if k1 else ""))
            traffic["fold_ablation_%d" % a] = {"what": nm, "cycles_per_launch": cyc, "mfma_busy_share": ga("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc),
                                               "valu_per_mfma": ga("SQ_INSTS_VALU") / ga("SQ_INSTS_MFMA"), "k1_run_line": k1[-1] if k1 else None}
    json.dump(traffic, open(os.path.join(outd, tag + "_match_traffic.json"), "w"), indent=1)
    open(os.path.join(outd, tag + "_match_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
    # kernel-trace summaries
    for name, sub in ((tag + "_bench_kernel_stats.csv", "trace_bench"), (tag + "_ba_cfg5_kernel_stats.csv", "trace_ba5"), (tag + "_ba_cfg4_kernel_stats.csv", "trace_ba4"), (tag + "_k1_cfg3_kernel_stats.csv", "trace_k1_cfg3")):
        st = kernel_stats(os.path.join(prof, sub))
        with open(os.path.join(outd, name), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats (tools/profile_round.sh, %s); source_hash %s git %s\n" % (sub, stamp["source_hash"], stamp.get("git_head", "?")[:12]))
            f.write("kernel,calls,avg_us,total_ms,percent\n")
            for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
                f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (k, v["calls"], v["avg_us"], v["total_ms"], v["pct"]))
    # BA chain counters
    sq, fe, wr = counters(os.path.join(prof, "ba5_sq")), counters(os.path.join(prof, "ba5_fetch")), counters(os.path.join(prof, "ba5_write"))
    bl = ["# BA cfg 5 (1000 cams / 100k pts / 1M obs), tools/ba_run.py 1000 100000 1, rocprofv3 --pmc passes; source_hash %s git %s" % (stamp["source_hash"], stamp.get("git_head", "?")[:12]),
          "# FETCH_SIZE / WRITE_SIZE in KiB per launch (FETCH to be doubled for wide streaming reads); MFMA share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)"]
    bl.append("# derived, per launch: fabric bytes = 1024 x (2 x FETCH_SIZE + WRITE_SIZE); MFMA-pipe busy share")
    for k in sorted(set(kk for (kk, c) in fe)):
        f, w = fe.get((k, "FETCH_SIZE"), (float("nan"), 0))[0], wr.get((k, "WRITE_SIZE"), (float("nan"), 0))[0]
        cyc = sq.get((k, "GRBM_GUI_ACTIVE"), (float("nan"), 0))[0] / 8.0
        busy = sq.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), (float("nan"), 0))[0]
        bl.append("%-44s fabric_bytes=%.4g  mfma_busy_share=%.3f  cycles=%.4g" % (k[:44], 1024.0 * (2.0 * f + w), busy / (1024.0 * cyc) if cyc == cyc and cyc > 0 else float("nan"), cyc))
    bl.append("# raw counter means")
    for (k, c), (v, n) in sorted(sq.items()):
        bl.append("%-44s %-28s n=%d mean=%.5g" % (k[:44], c, n, v))
    for tab in (fe, wr):
        for (k, c), (v, n) in sorted(tab.items()):
            bl.append("%-44s %-28s n=%d mean=%.5g" % (k[:44], c, n, v))
    open(os.path.join(outd, tag + "_ba_pmc_summary.txt"), "w").write("\n".join(bl) + "\n")
    print("wrote", outd)


if __name__ == "__main__":
    main()

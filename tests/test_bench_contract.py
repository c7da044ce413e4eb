"""CPU suite: the committed bench line (profiles/r03_bench_final.json, written by bench.py on the MI355X box) carries every
field of the driver's contract, its numbers are consistent with each other and with the committed rocprof summaries, and
bench.py still emits those keys."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_final.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["workload"].startswith("cfg3: 1000 images x 4096 keypoints x 256-d, 499500 image pairs")
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["host"]["nproc"] >= c["cores"]
    # value = pair-distances of the step / time of the step; the dominant kernel fits inside the step
    pd = 499500 * 4096 * 4096
    assert abs(d["value"] - pd / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert r["launch_ms"] + r["rerank_ms"] + r["unique_ms"] <= d["ms_per_step"]
    assert abs(r["achieved"] - 2 * 256 * pd / (r["launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 1e-6
    # the host lists of every step are what the tables hold: 8 bytes per match
    assert d["config"]["host_list_bytes_per_step"] == 8 * d["config"]["matches_found"]
    # configs[1] rides along, through the same path and with the tables left in HBM
    c2 = d["cfg2"]
    assert abs(c2["value"] - 4950 * 2048 * 2048 / (c2["ms_per_step"] * 1e-3)) / c2["value"] < 1e-6 and c2["value_tables_left_in_hbm"] >= c2["value"] * 0.98
    assert d["ba"]["cfg5"]["workload"].startswith("1000 cams / 100000 points")
    assert d["ba"]["cfg5"]["rms_diff_vs_cpu_px"] <= 1e-5 and d["ba"]["cfg5"]["iterations_equal_to_cpu"] is True
    assert d["ba"]["cfg4"]["rms_diff_vs_cpu_px"] <= 1e-5
    # round 3: parity of the timed runs themselves, the exchange's one host synchronisation, the factorisation's share of the peak
    assert d["config"]["equal_to_cpu"] is True and d["config"]["pairs_compared_with_cpu"] >= 512
    assert c2["equal_to_cpu"] is True and c2["parity"]["pairs_compared"] == 4950
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["host_syncs_per_exchange"] <= 1
    ch = d["ba"]["cfg5"]["cholesky_roofline"]
    assert ch["bound"] == "mfma" and abs(ch["frac"] - ch["achieved"] / ch["peak"]) < 1e-9 and ch["frac"] >= 0.40
    sr = d["ba"]["cfg5"]["stream_roofline"]
    assert sr["bound"] == "hbm" and sr["launch_us"] <= 80.0 and abs(sr["frac"] - sr["achieved"] / sr["peak"]) < 1e-9
    for leg in ("sift128", "orb32", "mixed_magnitudes", "cfg1"):
        assert leg in d, leg
    assert d["mixed_magnitudes"]["pairs_without_image_50_identical"] is True and d["cfg1"]["equal_to_cpu"] is True


def test_quoted_traffic_comes_from_a_profile_of_the_same_sources():
    """roofline.traffic is only ever the figure of a PMC profile taken on the sources the line's binary was built from."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_final.json")))
    t = json.load(open(os.path.join(ROOT, "profiles", "r03_match_traffic.json")))
    r = d["roofline"]
    if r["traffic"] is not None:
        assert r["source_hash"] == t["source_hash"]
        assert r["traffic"] == t["k_coarse_top2<256>@1000x4096"]["traffic_bytes_per_launch"]
        assert r["traffic"] > r["algorithmic_hbm_bytes_per_launch"]
    # every kernel a committed round-3 profile names is a kernel of the shipping sources
    src = "".join(open(os.path.join(ROOT, "reconstructor_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "reconstructor_amd", "csrc")) if f.endswith((".hip", ".h")))
    for name in ("r03_bench_kernel_stats.csv", "r03_ba_cfg5_kernel_stats.csv", "r03_ba_cfg4_kernel_stats.csv"):
        for ln in open(os.path.join(ROOT, "profiles", name)).read().splitlines()[2:]:
            k = ln.split(",")[0].strip('"').replace("void ", "")
            base = re.split(r"[<(]", k)[0]
            if base.startswith("_Z"):
                base = re.match(r"_Z\d+([A-Za-z_0-9]+?)(I|E|P|v)", base).group(1) if re.match(r"_Z\d+([A-Za-z_0-9]+?)(I|E|P|v)", base) else base
            if base.startswith("__amd_rocclr") or "nccl" in base.lower() or "rccl" in base.lower() or not base.startswith("k_"):
                continue
            assert re.search(r"\b%s\b" % re.escape(base), src), (name, k)


def test_kernel_trace_of_the_dominant_kernel_agrees_with_the_live_measurement():
    """profiles/r03_k1_cfg3_kernel_stats.csv is a rocprofv3 --kernel-trace --stats run in which every launch of the dominant
    kernel has the headline shape; its average must agree with the HIP-event time bench.py measured (another box of the pool:
    devices differ by a few per cent under this power-limited kernel)."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_final.json")))
    rows = [ln.split(",") for ln in open(os.path.join(ROOT, "profiles", "r03_k1_cfg3_kernel_stats.csv")).read().splitlines()[2:]]
    k1 = [r for r in rows if "k_coarse_top2<256" in r[0]]
    assert len(k1) == 1
    traced_ms = float(k1[0][-3]) / 1e3
    assert abs(traced_ms - d["roofline"]["launch_ms"]) / d["roofline"]["launch_ms"] < 0.08, (traced_ms, d["roofline"]["launch_ms"])


def test_bench_source_emits_every_contract_key():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(r'"%s"\s*[:\]]' % k, src), k
    assert "--gpus" in src and "--steps" in src and "--warmup" in src
    assert "from oracle" in src          # only in the cpu_baseline legs


def test_bench_refuses_a_world_it_cannot_build():
    """`python bench.py --gpus N` starts the N ranks itself; with fewer than N devices visible (none, in the build
    container) it exits non-zero with a message instead of running -- and reporting -- a smaller world."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"rccl"' in src and "torch.distributed.run" in src

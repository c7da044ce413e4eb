"""CPU suite: the committed bench line (profiles/r01_bench_final.json, written by bench.py on the MI355X
box) carries every field of the driver's contract, and bench.py still emits those keys."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_final.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value = pair-distances of the step / time of the step
    pd = 4950 * 2048 * 2048
    assert abs(d["value"] - pd / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["ba"]["cfg5"]["workload"].startswith("1000 cams / 100000 points")


def test_bench_source_emits_every_contract_key():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(r'"%s"\s*[:\]]' % k, src), k
    assert "--gpus" in src and "--steps" in src and "--warmup" in src
    assert "from oracle" in src          # only in the cpu_baseline legs

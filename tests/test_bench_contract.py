"""The one-line JSON contract of bench.py, checked on the committed line of the round's final GPU run (profiles/round2_bench_final.json):
required keys and types, the internal consistency of `roofline` (achieved = algorithmic FLOPs per launch / average launch time, frac =
achieved / peak, the PMC traffic file it cites exists and names the same kernel) and of `value` (images / s = 1000 / ms_per_step at N = 1)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_honours_the_contract():
    line = open(os.path.join(ROOT, "profiles", "round2_bench_final.json")).read().strip().split("\n")[-1]
    d = json.loads(line)
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["achieved"] - r["alg_flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) < 1e-3 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"].startswith("igemm_dma_kernel<") and r["traffic"] > r["traffic_alg_bytes_same_launches"] > 0
    pmc = json.load(open(os.path.join(ROOT, "profiles", "round2_pmc_traffic.json")))
    hit = [v for v in pmc["kernels"].values() if v.get("template") == r["kernel"]]
    assert len(hit) == 1 and hit[0]["traffic_bytes"] == r["traffic"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    for extra in ("pruned_schedule", "batched"):
        assert d[extra]["unit"] == "images/s" and d[extra]["value"] > 0

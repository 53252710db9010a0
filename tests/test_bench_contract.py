"""The one-line JSON contract of bench.py, checked on the line the driver's command produced on the round's final build -- the NEWEST
committed `profiles/round<N>_bench.json` (round 5's once it exists, round 4's before): required keys and types, `value` against
`ms_per_step`, the internal consistency of `roofline` (achieved = algorithmic FLOPs per launch / average launch time, frac = achieved /
peak, the counter summary it cites exists, carries the same kernel-source hash and holds the number quoted, the rocprofv3 figure is
recomputable from the committed CSV), of `phases` (the loop fraction from its own milliseconds) and of the extras."""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNET_GFLOP, TEXT_KV_GFLOP, PEAK = 803.27, 2.95, 2500.0      # bench.py's constants


def _newest_line():
    files = glob.glob(os.path.join(ROOT, "profiles", "round*_bench.json"))
    files = [f for f in files if re.fullmatch(r"round\d+_bench\.json", os.path.basename(f))]
    assert files, "no committed bench line under profiles/"
    path = max(files, key=lambda f: int(re.findall(r"\d+", os.path.basename(f))[0]))
    return path, json.loads(open(path).read().strip().split("\n")[-1])


def test_committed_bench_line_honours_the_contract():
    path, d = _newest_line()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict), ("phases", dict)):
        assert isinstance(d[k], t), (path, k)
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"]
    assert d["unit"] == "images/s" and d["data"] == "synthetic" and d["dtype"].startswith("f16")
    assert "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["config"]["unet_sample_forwards_per_image"] == 650.0 and d["config"]["ddim_steps"] == 50      # the faithful schedule, nothing skipped
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == PEAK
    assert abs(r["achieved"] - r["alg_flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) < 1e-3 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"].startswith("igemm_") and r["launches"] > 0 and r["alg_bytes_per_launch"] > 0
    assert r["launches"] * r["avg_launch_us"] * 1e-3 < d["ms_per_step"]          # the dominant kernel's launches fit inside one edit
    assert re.fullmatch(r"[0-9a-f]{16}", r["source_sha16"])
    if r.get("traffic") is not None:
        # per launch, from the committed counter summary of the same kernel sources
        src = r["traffic_source"].split(" ")[0]
        pmc = json.load(open(os.path.join(ROOT, src)))
        assert pmc["source_sha16"] == r["source_sha16"]
        hit = [v for v in pmc["kernels"].values() if v.get("template") == r["kernel"].replace(" ", "")]
        assert len(hit) == 1 and hit[0]["traffic_bytes"] == r["traffic"]
        assert abs(r["traffic_over_alg"] - r["traffic"] / r["alg_bytes_per_launch"]) < 1e-9
    else:
        assert "traffic_note" in r
    if "rocprof" in r:
        rp = r["rocprof"]
        stats = os.path.join(ROOT, rp["source"].split(":")[0])
        rows = [row for row in csv.DictReader(open(stats)) if abs(float(row["AverageNs"]) / 1e3 - rp["avg_launch_us"]) < 1e-6]
        assert rows and int(rows[0]["Calls"]) == rp["calls"]
        assert abs(rp["frac"] - r["alg_flop_per_launch"] / rp["avg_launch_us"] / 1e6 / PEAK) < 1e-9
    ph = d["phases"]
    step_ms = ph.get("lockstep_step_ms", ph.get("twelve_row_step_ms"))
    loop_flop = 50 * 12 * (UNET_GFLOP - TEXT_KV_GFLOP) * 1e9 + 12 * TEXT_KV_GFLOP * 1e9
    assert abs(ph["lockstep_loop_ms"] - 50 * step_ms) < 1e-6 * ph["lockstep_loop_ms"]
    assert abs(ph["lockstep_loop_mfma_frac"] - loop_flop / (ph["lockstep_loop_ms"] * 1e-3) / 1e12 / PEAK) < 1e-3 * ph["lockstep_loop_mfma_frac"]
    assert ph["ddim_inversion_ms"] + ph["lockstep_loop_ms"] < 1.05 * d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "images/s"
    for extra in ("pruned_schedule", "batched"):
        assert d[extra]["unit"] == "images/s" and d[extra]["value"] > 0
    b = d["batched"]
    if "phases" in b:       # round 5: the loop of BASELINE config 3's launch shape next to the single-image one
        bp, nb = b["phases"], b["images_per_launch_set_per_gpu"]
        assert bp["lockstep_rows"] == 12 * nb and bp["inversion_rows"] == nb
        lf = 50 * 12 * nb * (UNET_GFLOP - TEXT_KV_GFLOP) * 1e9 + 12 * nb * TEXT_KV_GFLOP * 1e9
        assert abs(bp["lockstep_loop_mfma_frac"] - lf / (bp["lockstep_loop_ms"] * 1e-3) / 1e12 / PEAK) < 1e-3 * bp["lockstep_loop_mfma_frac"]
    if "round6" in os.path.basename(path) or d.get("clock"):      # round 6: the clock the number was measured at travels with it
        ck = d["clock"]
        for probe in ("mfma_probe_before", "mfma_probe_after"):
            assert 0.5 < ck[probe]["effective_ghz"] <= 2.45 and ck[probe]["probe_ms"] > 1.0, ck[probe]
        dur = ck["during_timed_region"]
        assert "source" in dur and "samples" in dur
        if dur["source"] is not None:
            assert dur["samples"] >= 1 and 100.0 < dur["sclk_mhz"]["mean"] <= 2500.0 and dur["sclk_mhz"]["min"] <= dur["sclk_mhz"]["mean"] <= dur["sclk_mhz"]["max"]
        if "pipelined" in b:
            assert b["pipelined"].get("error") or (b["pipelined"]["unit"] == "images/s" and b["pipelined"]["value"] > 0)

#!/bin/bash
# The ping-pong GEMM kernel and its ablations on one conv shape: kernel-trace durations and one --pmc pass of SQ / GRBM counters (separate
# rocprofv3 runs), reduced to per-kernel averages incl. the effective shader clock (GRBM_GUI_ACTIVE cycles / kernel duration).
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/prof_pp"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$R"
CFG=${CFG:-17}
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python tools/pp_one.py $CFG 10 > "$OUT/kt.log" 2>&1
KT=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
pass() { local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" -d "$OUT/$name" -o $name --output-format csv -- python tools/pp_one.py $CFG 10 > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name "*counter_collection.csv" 2>/dev/null | head -1; }
A=$(pass c1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES)
B=$(pass c2 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY)
python - "$KT" "$A" "$B" "$OUT/pp_counters_cfg$CFG.json" <<'PY'
import collections, csv, json, sys
kt, out = {}, {}
for r in csv.DictReader(open(sys.argv[1])):
    if "igemm_pp" in r["Name"]: kt[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
for path in sys.argv[2:4]:
    if not path: continue
    d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "igemm_pp" not in k: continue
        d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in d:
        o = out.setdefault(k, dict(kt.get(k, {})))
        for c in d[k]: o[c] = d[k][c] / n[k][c]          # per launch
for k, o in sorted(out.items()):
    if "GRBM_GUI_ACTIVE" in o and "avg_us" in o: o["clock_mhz_est"] = o["GRBM_GUI_ACTIVE"] / o["avg_us"]
    wc = o.get("SQ_WAVE_CYCLES", 0) or 1
    o["frac_wait_any"] = o.get("SQ_WAIT_ANY", 0) / wc; o["frac_wait_inst"] = o.get("SQ_WAIT_INST_ANY", 0) / wc; o["frac_active"] = o.get("SQ_ACTIVE_INST_ANY", 0) / wc
    o["mfma_busy_over_wave_cycles_x4"] = o.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc)
    print(k[-40:], {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in o.items()})
json.dump(out, open(sys.argv[4], "w"), indent=1)
PY
rm -rf "$OUT/c1" "$OUT/c2" "$OUT/kt"

#!/bin/bash
# SQ stall breakdown of the GEMM kernels on 12-row UNet forwards (tools/fwd_only.py): where the wave cycles go.
#   WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES  (MI355X_MICROARCH.md, PMC slots)
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/prof_sq"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$R"
export ROWS=${ROWS:-12} N=2 WARM=1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$OUT/sq_counters.txt"
pass() {
  local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" -d "$OUT/$name" -o $name --output-format csv -- python tools/fwd_only.py > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name "*counter_collection.csv" 2>/dev/null | head -1
}
A=$(pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES)
if [ -z "$A" ]; then echo "pass 1 failed"; tail -5 "$OUT/sq1.log"; exit 0; fi
python - "$A" "$OUT/sq_summary.json" <<'PY'
import collections, csv, json, sys
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]; d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
out = {}
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    out[k] = {"launches": n[k], **{c: v[c] for c in v}, "frac_wait_any": v["SQ_WAIT_ANY"] / wc, "frac_wait_inst": v["SQ_WAIT_INST_ANY"] / wc,
              "frac_active": v["SQ_ACTIVE_INST_ANY"] / wc, "frac_wait_inst_lds": v["SQ_WAIT_INST_LDS"] / wc,
              "lds_conflict_over_active": v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)}
    print("%-64s n=%5d wait_any %.2f wait_inst %.2f active %.2f wait_lds %.3f ldsconf/ldsact %.3f mfma_busy/wavecyc(x4) %.3f" % (
        k[:64], n[k], out[k]["frac_wait_any"], out[k]["frac_wait_inst"], out[k]["frac_active"], out[k]["frac_wait_inst_lds"],
        out[k]["lds_conflict_over_active"], v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc)))
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
rm -rf "$OUT/sq1"

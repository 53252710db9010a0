#!/bin/bash
# Kernel census of the 12-row lock-step forward: rocprofv3 --kernel-trace --stats of tools/fwd_only.py (ROWS forwards and nothing else).
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/prof_fwd"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$R"
export ROWS=${ROWS:-12} N=${N:-10} WARM=2
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python tools/fwd_only.py > "$OUT/fwd_only.log" 2>&1
tail -2 "$OUT/fwd_only.log"
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/fwd${ROWS}_kernel_stats.csv"
python - "$OUT/fwd${ROWS}_kernel_stats.csv" $((N + WARM)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); nf = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per forward: %.3f ms, launches per forward: %.0f" % (tot / nf / 1e6, sum(int(r["Calls"]) for r in rows) / nf))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%6.1f us/fwd  %5.1f calls/fwd  avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / nf / 1e3, int(r["Calls"]) / nf, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
rm -rf "$OUT/kt"

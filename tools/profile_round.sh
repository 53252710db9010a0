#!/bin/bash
# Round profiles on a GPU box (outputs under gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GPU active) of a short bench run, reduced by tools/pmc_summary.py
TAG="${1:-round2}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
set -x
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python bench.py --steps 2 --warmup 1 > "$OUT/bench_under_trace.json" 2> "$OUT/kt.err"
tail -c 600 "$OUT/bench_under_trace.json"
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
head -12 "$OUT/kernel_stats.csv"
SHORT="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 500 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- $SHORT > "$OUT/fetch.log" 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- $SHORT > "$OUT/write.log" 2>&1
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/mfma" -o mfma --output-format csv -- $SHORT > "$OUT/mfma.log" 2>&1
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1); W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1); M=$(find "$OUT/mfma" -name "*counter_collection.csv" | head -1)
ls -la "$F" "$W" "$M"
python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_traffic.json" "$M" | head -14
# the raw counter CSVs are large: keep only the summaries for the merge back
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/mfma"
find "$OUT/kt" -name "*kernel_trace.csv" -delete
du -sh "$OUT"

#!/bin/bash
# Round profiles on a GPU box (outputs under gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench command without its extras (only faithful edits: the per-kernel averages are those of the timed edits)
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GPU active) of a short bench run, reduced by tools/pmc_summary.py
TAG="${1:-round2}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
set -x
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > "$OUT/bench_under_trace.json" 2> "$OUT/kt.err"
tail -c 600 "$OUT/bench_under_trace.json"
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
python -c "import json; from pnpinversion_amd.build import source_hash; json.dump({'source_sha16': source_hash(), 'command': 'rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline'}, open('$OUT/kernel_stats.meta.json', 'w'))"
head -12 "$OUT/kernel_stats.csv"
# BASELINE config 3's launch shape (8 images per set of launches: 8-row inversion, 96-row lock-step forwards) under the same tracer
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/kt96" -o kt96 --output-format csv -- python tools/batched_only.py 8 50 1 > "$OUT/batched_under_trace.json" 2> "$OUT/kt96.err"
tail -c 300 "$OUT/batched_under_trace.json"
find "$OUT/kt96" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/batched8_kernel_stats.csv"
find "$OUT/kt96" -name "*kernel_trace.csv" -delete
# The counter passes run the SAME command with two DDIM steps instead of fifty: rocprofv3's counter service segfaults inside its dispatch
# interception once a process has launched some tens of thousands of kernels (a full edit is ~50 000 launches, plus the event-bracketed
# profiling edit; tools/fwd_only.py with 1 600 never tripped it).  Same kernels, same launch shapes, same 1 : 1 mix of one-row and
# twelve-row forwards -- per-launch averages are what the summary keeps.
SHORT="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --ddim-steps 2"
export PMC_SOURCE="rocprofv3 --pmc passes (one counter set per run) of: $SHORT"
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

#!/bin/bash
# PMC passes on 12-row UNet forwards (tools/fwd_only.py): FETCH_SIZE | WRITE_SIZE | MFMA busy, each in its own rocprofv3 run.
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/prof_pmc"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$R"
export ROWS=12 N=2 WARM=1
pass() {  # name, counters...
  local name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" -d "$OUT/$name" -o $name --output-format csv -- python tools/fwd_only.py > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name "*counter_collection.csv" 2>/dev/null | head -1
}
F=$(pass fetch FETCH_SIZE)
if [ -z "$F" ]; then
  echo "default configuration crashed under --pmc; retrying without the deep LDS rings"; tail -3 "$OUT/fetch.log"
  export TUNE="igemm_deep_rings=0"
  F=$(pass fetch FETCH_SIZE)
fi
if [ -z "$F" ]; then echo "PMC collection failed"; tail -5 "$OUT/fetch.log"; exit 0; fi
echo "TUNE=$TUNE" > "$OUT/config.txt"
W=$(pass write WRITE_SIZE); M=$(pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)
python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_traffic.json" $M | head -12
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/mfma"

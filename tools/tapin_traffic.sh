#!/bin/bash
# Fabric traffic and duration of the dominant 3 x 3 convolution (12 rows, 64 x 64, 320 -> 320, igemm_pp<192,320>) under the tap-major K walk
# (default) and the channel-slab-major one (tuning igemm_tapin = 1): FETCH_SIZE per launch (x2: gfx950 correction) and kernel-trace average.
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/prof_tapin"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$R"
export VPP=0
for T in 0 1; do
  export PNPI_TUNE="igemm_tapin=$T"
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/kt$T" -o kt --output-format csv -- python tools/pp_one.py 17 20 > "$OUT/kt$T.log" 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/f$T" -o f --output-format csv -- python tools/pp_one.py 17 20 > "$OUT/f$T.log" 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/w$T" -o w --output-format csv -- python tools/pp_one.py 17 20 > "$OUT/w$T.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = {}
for T in (0, 1):
    o = {}
    for f in glob.glob("%s/kt%d/**/*kernel_stats.csv" % (sys.argv[1], T), recursive=True):
        for r in csv.DictReader(open(f)):
            if "igemm_pp" in r["Name"]: o["avg_us"] = float(r["AverageNs"]) / 1e3; o["calls"] = int(r["Calls"])
    for key, d in (("fetch", "f"), ("write", "w")):
        for f in glob.glob("%s/%s%d/**/*counter_collection.csv" % (sys.argv[1], d, T), recursive=True):
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "igemm_pp" in r["Kernel_Name"]]
            if v: o[key + "_counter_per_launch"] = sum(v) / len(v)
    out["tapin_%d" % T] = o
json.dump(out, open(sys.argv[1] + "/tapin_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf "$OUT"/kt? "$OUT"/f? "$OUT"/w?

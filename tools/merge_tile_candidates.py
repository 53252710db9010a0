"""Patch csrc/tile_table.inc with NEW tile configurations measured after the table was generated: each input is a tools/fwd_tune.py
JSON of a partial sweep (FWD_TUNE_CFGS=...: `auto` = what the current table runs for the shape + the new configurations, all timed in
the same process on the same box); an entry is replaced when a new configuration beats `auto` by more than --margin (default 3 %:
interleaved arms of one process repeat to about 1 - 2 %).  Entries the sweeps did not see, or did not beat, stay as they are.
usage: merge_tile_candidates.py [--margin 0.03] table.inc sweep.json [sweep.json ...]"""
import json, re, sys
CFG = {"128": 0, "64": 1, "256m": 3, "320": 4, "256n": 5, "256x320": 6, "256x256": 7, "64k4": 8, "128k2": 9, "128k2b": 10, "64k2": 11, "64x320": 12, "128n2": 13, "128b64": 14,
       "256nb64": 15, "pp256": 16, "pp320": 17, "128b64x3": 18}      # tools/gen_tile_table.py's names
args = sys.argv[1:]
margin = 0.03
if args[0] == "--margin":
    margin = float(args[1]); args = args[2:]
table, sweeps = args[0], args[1:]
best = {}
for path in sweeps:
    d = json.load(open(path))
    for sh in d["shapes"]:
        auto = sh["us"].get("auto")
        if not auto:
            continue
        cands = {k: v for k, v in sh["us"].items() if k != "auto" and k.partition("/s")[0] in CFG}
        if not cands:
            continue
        key, us = min(cands.items(), key=lambda kv: kv[1])
        if us < (1.0 - margin) * auto:
            base, _, sp = key.partition("/s")
            k = (sh["M"], sh["N"], sh["K"], sh["ks"])
            if k not in best or us / auto < best[k][3]:
                best[k] = (CFG[base], int(sp) if sp else 1, us, us / auto, auto, d["rows"])
out, n = [], 0
for line in open(table):
    m = re.match(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},", line)
    if m:
        k = tuple(int(x) for x in m.groups()[:4])
        if k in best:
            cfg, split, us, _, auto, rows = best[k]
            line = "{%d, %d, %d, %d, %d, %d},   // %d rows: %.1f us (was %.1f us under {%s, %s}; tools/merge_tile_candidates.py)\n" % (k + (cfg, split, rows, us, auto, m.group(5), m.group(6)))
            n += 1
    out.append(line)
open(table, "w").writelines(out)
print("%d of %d candidate shapes replaced in %s" % (n, len(best), table))
for k, v in sorted(best.items()):
    print("  %s -> cfg %d split %d: %.1f us against %.1f" % (k, v[0], v[1], v[2], v[4]))

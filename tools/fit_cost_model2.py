"""Offline refit of launch_igemm's cost model against tools/autotune2.py output (gpurun_out/autotune_r2_b*.json).
model:  t = t_launch + ceil(units / 256) * (t_fix[tile] + chunks_per_unit * 2*bm*bn*64 / (rate[tile] / 256)) / resid(per_cu, bpc[tile])
            + [split > 1] * ((2*s + 1) * M*N*4 / bw_slab + t_reduce)
usage: fit_cost_model2.py a.json [b.json ...]"""
import json, math, os, sys
import numpy as np
from scipy.optimize import least_squares

TILES = {"128": (128, 128, 3), "64": (64, 64, 4), "320": (128, 320, 2), "256n": (128, 256, 2)}      # default variants (v1)
if os.environ.get("FIT_ALL_TILES"):
    TILES.update({"128k2b": (128, 128, 1), "64k2": (64, 64, 2), "256x256": (256, 256, 1), "256x320": (256, 320, 1)})
names = list(TILES)
rows = []
for path in sys.argv[1:]:
    d = json.load(open(path))
    for sh in d["shapes"]:
        M, N, K = sh["M"], sh["N"], sh["K"]
        for key, us in sh["us"].items():
            if key == "auto": continue
            base, _, sp = key.partition("/s")
            s = int(sp) if sp else 1
            if os.environ.get("FIT_R2A") and s == 1:      # the first round-2 run: default variant was v0, "v1" is today's default
                base = {"320v1": "320", "256nv1": "256n", "320": "320v0", "256n": "256nv0"}.get(base, base)
            if base not in TILES: continue
            rows.append((M, N, K, base, s, us))
print(len(rows), "measurements")

def predict(p, M, N, K, base, s):
    i = names.index(base)
    bm, bn, bpc = TILES[base]
    rate, tfix = p[2 * i] * 1e12, p[2 * i + 1] * 1e-6
    NT = 2 * len(names)
    t_launch, r1, bw, t_red = p[NT] * 1e-6, p[NT + 1], p[NT + 2] * 1e12, p[NT + 3] * 1e-6
    nch = (K + 63) // 64
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    units = tiles * s
    per_cu = units / 256.0
    fill = 1.0 if per_cu >= bpc else (0.0 if per_cu <= 1.0 else (per_cu - 1.0) / (bpc - 1.0))
    resid = r1 + (1.0 - r1) * fill
    unit = tfix + math.ceil(nch / s) * 2.0 * bm * bn * 64 / (rate / 256.0)
    t = t_launch + math.ceil(units / 256.0) * unit / resid
    if s > 1: t += (2 * s + 1) * M * N * 4.0 / bw + t_red
    return t * 1e6

def resid_fn(p):
    return [math.log(predict(p, M, N, K, b, s) / us) for (M, N, K, b, s, us) in rows]

p0 = [840, 1.0] * len(names) + [3.0, 0.7, 2.5, 8.0]
lo = [200, 0] * len(names) + [0, 0.3, 0.5, 0]
hi = [3000, 30] * len(names) + [20, 1.0, 12, 50]
sol = least_squares(resid_fn, p0, bounds=(lo, hi))
p = sol.x
r = np.array(resid_fn(p))
print("rms log error %.3f" % math.sqrt((r ** 2).mean()))
for i, n in enumerate(names):
    print("  %-7s rate %.0f TF  t_fix %.2f us" % (n, p[2 * i], p[2 * i + 1]))
NT = 2 * len(names)
print("  t_launch %.2f us  resid(1 block/CU) %.2f  slab bw %.2f TB/s  t_reduce %.2f us" % (p[NT], p[NT + 1], p[NT + 2], p[NT + 3]))
# regret of the model's choice per shape
tot_pick = tot_best = 0.0
for path in sys.argv[1:]:
    d = json.load(open(path))
    for sh in d["shapes"]:
        M, N, K = sh["M"], sh["N"], sh["K"]
        cands = {}
        for key, us in sh["us"].items():
            base, _, sp = key.partition("/s")
            s = int(sp) if sp else 1
            b2 = base
            if os.environ.get("FIT_R2A") and s == 1: b2 = {"320v1": "320", "256nv1": "256n", "320": "320v0", "256n": "256nv0"}.get(base, base)
            if b2 in TILES: cands[key] = (predict(p, M, N, K, b2, s), us)
        pick = min(cands, key=lambda k: cands[k][0])
        best = min(cands, key=lambda k: cands[k][1])
        tot_pick += cands[pick][1] * sh["n_per_fwd"]; tot_best += cands[best][1] * sh["n_per_fwd"]
        if cands[pick][1] > 1.05 * cands[best][1]:
            print("  regret M=%d N=%d K=%d: pick %s %.0f us, best %s %.0f us" % (M, N, K, pick, cands[pick][1], best, cands[best][1]))
print("per forward: model's picks %.2f ms, per-shape best %.2f ms" % (tot_pick / 1e3, tot_best / 1e3))
print("params:", ", ".join("%.4g" % x for x in p))

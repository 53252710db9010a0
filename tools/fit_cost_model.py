"""Offline refit of launch_igemm's tile / split-K cost model against tools/autotune_report.py tables (gpurun_out/autotune_b*.txt).
Evaluates the regret (time of the model's pick minus the best measured configuration, weighted by launches per forward)."""
import glob, itertools, re

def parse(fn):
    recs = []
    for line in open(fn):
        m = re.match(r"M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) ks=(\d)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+best=\S+.*\| (.*)$", line)
        if not m: continue
        M, N, K, ks = map(int, m.group(1, 2, 3, 4))
        cfgs = {}
        for tok in m.group(8).split():
            k, v = tok.split(":"); cfgs[k] = float(v)
        recs.append(dict(M=M, N=N, K=K, ks=ks, n=float(m.group(5)), auto=float(m.group(6)), cfgs=cfgs))
    return recs

def model_pick(r, P):
    M, N, K = r["M"], r["N"], r["K"]
    nchunks = (K + 63) // 64
    best, pick = 1e30, None
    for tile in (128, 64):
        tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
        rate = P["r128"] if tile == 128 else P["r64"]
        flops = 2.0 * tiles * tile * tile * K
        for s in ((1, 2, 3, 4) if tile == 128 else (1, 2, 3, 4, 6, 8, 12)):
            if s > 1 and nchunks // s < P["minchunks"]: continue
            u = tiles * s / 256.0
            busy = min(u, 1.0)
            xs, ys = P["knots"]
            uu = max(u, 1.0)
            res = ys[-1]
            for i in range(len(xs) - 1):
                if uu < xs[i + 1]:
                    res = ys[i] + (ys[i + 1] - ys[i]) * (uu - xs[i]) / (xs[i + 1] - xs[i]); break
            if u < 1.0: res = ys[0]
            t = flops / (rate * busy * res)
            # short-K launches pay a fixed prologue / epilogue cost per tile wave
            t += P["fix"] * 1e-6
            if s > 1: t += (2 * s + 1) * M * N * 4.0 / P["redbw"] + P["redfix"] * 1e-6
            if t < best: best, pick = t, ("%d" % tile if s == 1 else "%d/s%d" % (tile, s))
    return pick

def regret(tables, P, verbose=False):
    tot = base = bestsum = 0.0
    for name, recs in tables.items():
        for r in recs:
            pick = model_pick(r, P)
            if pick not in r["cfgs"]:
                # unmeasured combination: approximate with the nearest measured split of that tile
                cands = [k for k in r["cfgs"] if k.split("/")[0] == pick.split("/")[0]]
                pick = min(cands, key=lambda k: abs(int((k.split("/s") + ["1"])[1]) - int((pick.split("/s") + ["1"])[1])))
            t = r["cfgs"][pick]; b = min(r["cfgs"].values())
            tot += r["n"] * t; bestsum += r["n"] * b; base += r["n"] * r["auto"]
            if verbose and t > 1.06 * b:
                print("  %s M=%d N=%d K=%d: pick %s %.0f vs best %.0f (n=%.0f)" % (name, r["M"], r["N"], r["K"], pick, t, b, r["n"]))
    return tot, base, bestsum

if __name__ == "__main__":
    tables = {fn.split("_")[-1][:-4]: parse(fn) for fn in sorted(glob.glob("gpurun_out/autotune_b*.txt"))}
    cur = dict(r128=720e12, r64=460e12, knots=([1, 2, 3], [0.55, 0.85, 1.0]), redbw=2.5e12, redfix=4.0, fix=0.0, minchunks=6)
    print("current model:", ["%.2f" % (x / 1e3) for x in regret(tables, cur)], "(model, measured-auto, per-shape best) ms")
    best = (1e30, None)
    for r128, r64, y0, y1, rb, rf, fix, mc in itertools.product((760e12, 800e12, 840e12), (520e12, 560e12, 600e12), (0.6, 0.7, 0.8), (0.85, 0.92, 0.97),
                                                                (1.5e12, 2.5e12, 4e12), (3.0, 6.0, 10.0), (0.0, 3.0), (4, 6)):
        P = dict(r128=r128, r64=r64, knots=([1, 2, 3], [y0, y1, 1.0]), redbw=rb, redfix=rf, fix=fix, minchunks=mc)
        t = regret(tables, P)[0]
        if t < best[0]: best = (t, P)
    print("best:", "%.2f ms" % (best[0] / 1e3), best[1])
    regret(tables, best[1], verbose=True)
    for name in tables:
        sub = {name: tables[name]}
        print(name, ["%.2f" % (x / 1e3) for x in regret(sub, best[1])])

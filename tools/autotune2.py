"""Every tile / split-K configuration (and kernel variant) of launch_igemm per distinct GEMM / conv shape of a UNet forward, cold
weights as in a forward.  Shapes come from a PNPI_PROFILE_DUMP csv (profiles/round*_launch_dump_b<rows>.csv).
usage: autotune2.py dump.csv rows out.json [min_share]"""
import collections, csv, json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"; lib = ctx.lib
rows_b = int(sys.argv[2]); out_path = sys.argv[3]
min_share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.003
full = os.environ.get("AUTOTUNE_FULL", "0") == "1"    # also the non-default kernel variants
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["cls"]) not in (0, 1, 2, 9): continue
    k = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["ksize"]))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["us"])
nfwd = 3
tot_us = sum(v[1] for v in agg.values())

def timeit(fn, ncopy):
    for i in range(ncopy): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(3 * ncopy, 12)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def tune(**kw):
    for k, v in kw.items():
        assert lib.pnpi_set_tuning(k.encode(), v) == 0

res_all = []
for (M, N, K, ks), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / tot_us < min_share: continue
    ncopy = min(16, max(2, int(300e6 / (N * K * 2)) + 1))     # rotate weights through > 256 MB where they are large (cold, as in a forward)
    ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).half() for _ in range(ncopy)]
    if ks == 3:
        cin = K // 9; hw = int(round(math.sqrt(M // rows_b)))
        if hw * hw * rows_b != M or cin % 8: continue
        x = torch.randn(rows_b, hw, hw, cin, device=DEV).half(); out = torch.empty(rows_b, hw, hw, N, device=DEV, dtype=torch.half)
        bias = torch.randn(N, device=DEV)
        def mk(cfg, sp):
            return lambda i: ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, rows_b, hw, hw, 3, 1, 1, 0, hw, hw, ptr(ws[i % ncopy]), ptr(bias), None, N, ptr(out), cfg, sp)
    else:
        a = torch.randn(M, K, device=DEV).half(); o = torch.empty(M, N, device=DEV, dtype=torch.half)
        def mk(cfg, sp):
            return lambda i: ctx.call("pnpi_op_gemm", ptr(a), K, ptr(ws[i % ncopy]), K, M, N, K, 1.0, None, None, ptr(o), N, 1 << 30, None, 0, 0, 1, cfg, sp)
    res = {}
    res["auto"] = timeit(mk(-1, 0), ncopy)
    res["128"] = timeit(mk(0, 0), ncopy)
    res["64"] = timeit(mk(1, 0), ncopy)
    fast = (K // (ks * ks)) % 64 == 0
    if fast:
        res["320"] = timeit(mk(4, 0), ncopy)
        res["256n"] = timeit(mk(5, 0), ncopy)
        res["256x320"] = timeit(mk(6, 0), ncopy)
        res["256x256"] = timeit(mk(7, 0), ncopy)
        if full:
            for v in (0, 2):
                tune(igemm_v320=v); res["320v%d" % v] = timeit(mk(4, 0), ncopy); tune(igemm_v320=1)
                tune(igemm_v256n=v); res["256nv%d" % v] = timeit(mk(5, 0), ncopy); tune(igemm_v256n=1)
    if fast:
        for name, c in (("64k4", 8), ("64k2", 11), ("128k2", 9), ("128k2b", 10), ("64x320", 12), ("128n2", 13)):
            res[name] = timeit(mk(c, 0), ncopy)
    nch = K // 64
    for sp in (2, 3, 4, 6, 8, 12, 16):
        if nch // sp >= 4 and sp * M * N * 4 <= 96 << 20: res["64/s%d" % sp] = timeit(mk(2, sp), ncopy)
    for sp in (2, 3, 4, 6, 8):
        if nch // sp >= 4 and sp * M * N * 4 <= 96 << 20:
            res["128/s%d" % sp] = timeit(mk(0, sp), ncopy)
            if fast:
                res["320/s%d" % sp] = timeit(mk(4, sp), ncopy)
                res["256n/s%d" % sp] = timeit(mk(5, sp), ncopy)
                if sp <= 4 and nch // sp >= 8:
                    res["64k4/s%d" % sp] = timeit(mk(8, sp), ncopy)
                    res["128k2/s%d" % sp] = timeit(mk(9, sp), ncopy)
    best = min((v, k) for k, v in res.items() if k != "auto")
    n_per = cnt / nfwd
    fl = 2.0 * M * N * K
    res_all.append({"M": M, "N": N, "K": K, "ks": ks, "n_per_fwd": n_per, "dump_us": us / cnt, "us": res, "best": best[1]})
    top = sorted(((v, k) for k, v in res.items()), key=lambda t: t[0])[:6]
    print("M=%6d N=%5d K=%6d ks=%d n=%4.1f auto %7.1f (%4.0f TF) best %-9s %7.1f (%4.0f TF) | %s" % (
        M, N, K, ks, n_per, res["auto"], fl / res["auto"] / 1e6, best[1], best[0], fl / best[0] / 1e6,
        " ".join("%s:%.0f" % (k, v) for v, k in top)), flush=True)
    del ws
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump({"rows": rows_b, "shapes": res_all}, open(out_path, "w"), indent=0)
ta = sum(r["us"]["auto"] * r["n_per_fwd"] for r in res_all); tb = sum(min(v for k, v in r["us"].items() if k != "auto") * r["n_per_fwd"] for r in res_all)
t_old = sum(min(v for k, v in r["us"].items() if k != "auto" and not k.startswith(("320", "256", "64k", "128k", "64x", "128n"))) * r["n_per_fwd"] for r in res_all)
print("sum over listed shapes per forward: auto %.2f ms, per-shape best %.2f ms, best without the wide tiles %.2f ms" % (ta / 1e3, tb / 1e3, t_old / 1e3))

"""For every distinct GEMM / conv shape of a UNet forward (from a PNPI_PROFILE_DUMP csv) time the tile / split-K configurations
the launcher could pick (cold weights, as in a forward) and compare with what the cost model picked.
usage: autotune_report.py dump.csv rows"""
import collections, csv, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"
rows_b = int(sys.argv[2])
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["cls"]) > 2: continue
    k = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["ksize"]))
    a = agg.setdefault(k, [0, 0.0, int(r["cls"])]); a[0] += 1; a[1] += float(r["us"])
nfwd = 3
def timeit(fn, ncopy):
    for i in range(ncopy): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(3 * ncopy, 12)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot_auto = tot_best = 0.0
print("%-34s %5s %9s %9s  %s" % ("shape", "n/fwd", "auto us", "best us", "configs (cfg/split: us)"))
for (M, N, K, ks), (cnt, us, cls) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / sum(v[1] for v in agg.values()) < 0.004: continue
    ncopy = max(2, int(300e6 / (N * K * 2)) + 1)
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
    for sp in (2, 3, 4, 6, 8, 12):
        if K // 64 // sp >= 4: res["64/s%d" % sp] = timeit(mk(2, sp), ncopy)
    for sp in (2, 3, 4):
        if K // 64 // sp >= 6: res["128/s%d" % sp] = timeit(mk(0, sp), ncopy)
    best = min((v, k) for k, v in res.items() if k != "auto")
    n_per = cnt / nfwd
    tot_auto += res["auto"] * n_per; tot_best += best[0] * n_per
    flag = "  <-- %.0f%%" % (100 * (res["auto"] / best[0] - 1)) if res["auto"] > 1.06 * best[0] else ""
    print("M=%6d N=%5d K=%6d ks=%d %5.1f %9.1f %9.1f  best=%s%s | %s" % (M, N, K, ks, n_per, res["auto"], best[0], best[1], flag,
          " ".join("%s:%.0f" % (k, v) for k, v in res.items() if k != "auto")))
    del ws
print("sum over listed shapes per forward: auto %.2f ms, per-shape best %.2f ms" % (tot_auto / 1e3, tot_best / 1e3))

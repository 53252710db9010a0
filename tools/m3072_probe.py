"""Where the time of the 16^2 / 8^2-level layers of a 12-row forward goes (M = 3072 / 768: one wave of 128-row tiles): each shape with
COLD operands (activation and weight rotated through > 400 MB of copies, as inside a forward) under the table's choice, forced
configurations, and the 128 x 128 kernel's ablations (tuning igemm_v128 = 11 DMA only / 12 MFMAs only -- garbage results, timing only).
usage: m3072_probe.py        -> gpurun_out/m3072_probe.json"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"
NAME = {-1: "table", 0: "128", 1: "64", 10: "128k2b", 14: "128b64", 16: "pp256", 17: "pp320"}


def bench(go, n):
    for i in range(n): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


out = []
for (M, N, K) in [(3072, 1280, 1280), (3072, 1280, 5120), (3072, 3840, 1280), (768, 1280, 1280), (768, 1280, 5120)]:
    nc = max(2, int(420e6 / (N * K * 2)) + 1)
    Ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).half() for _ in range(nc)]
    As = [torch.randn(M, K, device=DEV).half() for _ in range(8)]
    bias = torch.randn(N, device=DEV); O = torch.empty(M, N, device=DEV, dtype=torch.half)
    rec = {"M": M, "N": N, "K": K, "us": {}}

    def run(cfg, split, label, v128=None):
        if v128 is not None:
            ctx.lib.pnpi_set_tuning(b"igemm_v128", v128)
        go = lambda i: ctx.call("pnpi_op_gemm", ptr(As[i % 8]), K, ptr(Ws[i % nc]), K, M, N, K, 1.0, ptr(bias), None, ptr(O), N, 1 << 30, None, 0, 0, 1, cfg, split)
        try:
            rec["us"][label] = round(bench(go, max(2 * nc, 40)), 2)
        except Exception as e:                      # a configuration the shape does not admit
            rec["us"][label] = None
        if v128 is not None:
            ctx.lib.pnpi_set_tuning(b"igemm_v128", 2)

    for cfg in (-1, 0, 10, 14, 1, 16, 17):
        run(cfg, 0, NAME[cfg])
    for s in (2, 4):
        run(0, s, "128/s%d" % s); run(16, s, "pp256/s%d" % s)
    run(0, 0, "128 ring 4 (v128=3)", 3); run(0, 0, "128 ring 8 (v128=8)", 8); run(0, 0, "128 K-tile 64 x 3 (v128=1)", 1)
    run(0, 0, "128 DMA only", 11); run(0, 0, "128 MFMAs only", 12)
    # the floor of a dependent launch in this stream: the same call on a 64 x 64 x 64 problem
    a0 = torch.randn(64, 64, device=DEV).half(); w0 = torch.randn(64, 64, device=DEV).half(); o0 = torch.empty(64, 64, device=DEV, dtype=torch.half)
    rec["us"]["launch floor (64^3)"] = round(bench(lambda i: ctx.call("pnpi_op_gemm", ptr(a0), 64, ptr(w0), 64, 64, 64, 64, 1.0, None, None, ptr(o0), 64, 1 << 30, None, 0, 0, 1, -1, 0), 200), 2)
    fl = 2.0 * M * N * K
    rec["ideal_us"] = {"mfma_2.5PF": round(fl / 2.5e15 * 1e6, 2), "hbm_8TB": round(2.0 * (M * K + N * K + M * N) / 8e12 * 1e6, 2)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/m3072_probe.json", "w"), indent=1)

"""Where the time of the short-K linear layers of a 12-row forward goes (K = 320 / 640 at 49 152 / 12 288 rows: byte-bound shapes that run at
0.2 - 0.4 of their byte bound): cold activations (rotated through eight copies), with and without the residual, under the table's choice, the
forced ping-pong / 4-wave configurations and the ping-pong kernel's ablations (tuning igemm_vpp = 1 no MFMAs / 2 no DMA / 3 DMA only /
4 MFMAs only -- garbage results, timing only), plus a device-to-device copy of the same bytes as the streaming floor.
usage: shortk12_probe.py        -> gpurun_out/shortk12_probe.json"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"
NAME = {-1: "table", 0: "128", 4: "320", 14: "128b64", 16: "pp256", 17: "pp320"}


def bench(go, n=48):
    for i in range(n): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


out = []
for (M, N, K) in [(49152, 320, 320), (12288, 640, 640), (49152, 1536, 320), (49152, 320, 1280)]:
    W = (torch.randn(N, K, device=DEV) / math.sqrt(K)).half()
    As = [torch.randn(M, K, device=DEV).half() for _ in range(8)]
    Rs = [torch.randn(M, N, device=DEV).half() for _ in range(8)]
    Os = [torch.empty(M, N, device=DEV, dtype=torch.half) for _ in range(8)]
    bias = torch.randn(N, device=DEV)
    rec = {"M": M, "N": N, "K": K, "us": {}}

    def run(cfg, label, res, vpp=0):
        ctx.lib.pnpi_set_tuning(b"igemm_vpp", vpp)
        go = lambda i: ctx.call("pnpi_op_gemm", ptr(As[i % 8]), K, ptr(W), K, M, N, K, 1.0, ptr(bias), ptr(Rs[i % 8]) if res else None, ptr(Os[i % 8]), N,
                                1 << 30, None, 0, 0, 1, cfg, 0)
        try:
            rec["us"][label + (" +res" if res else "")] = round(bench(go), 2)
        except Exception:
            rec["us"][label + (" +res" if res else "")] = None
        ctx.lib.pnpi_set_tuning(b"igemm_vpp", 0)

    for res in (0, 1):
        for cfg in (-1, 17, 16, 0, 14, 4):
            run(cfg, NAME[cfg], res)
    for vpp, lab in ((1, "pp320 no MFMAs"), (2, "pp320 no DMA"), (3, "pp320 DMA only"), (4, "pp320 MFMAs only")):
        run(17, lab, 0, vpp)
    # streaming floors: the bytes of the activation in and the output out (and the residual) as plain copies / adds
    rec["us"]["copy A -> O bytes (torch)"] = round(bench(lambda i: Os[i % 8].view(-1)[: min(M * K, M * N)].copy_(As[i % 8].view(-1)[: min(M * K, M * N)])), 2)
    rec["us"]["O = R + R' (torch add)"] = round(bench(lambda i: torch.add(Rs[i % 8], Rs[(i + 1) % 8], out=Os[i % 8])), 2)
    a0 = torch.randn(64, 64, device=DEV).half(); w0 = torch.randn(64, 64, device=DEV).half(); o0 = torch.empty(64, 64, device=DEV, dtype=torch.half)
    rec["us"]["launch floor (64^3)"] = round(bench(lambda i: ctx.call("pnpi_op_gemm", ptr(a0), 64, ptr(w0), 64, 64, 64, 64, 1.0, None, None, ptr(o0), 64, 1 << 30, None, 0, 0, 1, -1, 0), 200), 2)
    byt = 2.0 * (M * K + N * K + M * N)
    rec["ideal_us"] = {"mfma_2.5PF": round(2.0 * M * N * K / 2.5e15 * 1e6, 2), "hbm_8TB": round(byt / 8e12 * 1e6, 2), "hbm_8TB +res": round((byt + 2.0 * M * N) / 8e12 * 1e6, 2)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/shortk12_probe.json", "w"), indent=1)

"""All five forms of the ping-pong kernel (tuning igemm_vpp: 0 full, 1 no MFMAs, 2 no DMA, 3 DMA only, 4 MFMAs only; 1 - 4 produce garbage) on GEMM
shapes with cold operands, both geometries: which pair of the three streams (DMA, fragment reads, MFMAs) costs the time.
usage: pp_ablate.py  -> gpurun_out/pp_ablate.json"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"


def bench(go, n=24):
    for i in range(n): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


out = []
for (M, N, K) in [(12288, 1280, 11520), (49152, 640, 5760), (49152, 320, 2880)]:
    Ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).half() for _ in range(6)]
    As = [torch.randn(M, K, device=DEV).half() for _ in range(2)]
    O = torch.empty(M, N, device=DEV, dtype=torch.half); bias = torch.randn(N, device=DEV)
    for cfg in (16, 17):
        rec = {"M": M, "N": N, "K": K, "cfg": cfg, "mfma_us_at_2.5PF": round(2.0 * M * N * K / 2.5e15 * 1e6, 1)}
        for vpp, lab in ((0, "full"), (1, "DMA + reads (no MFMAs)"), (2, "MFMAs + reads (no DMA)"), (3, "DMA only"), (4, "MFMAs only")):
            assert ctx.lib.pnpi_set_tuning(b"igemm_vpp", vpp) == 0
            go = lambda i: ctx.call("pnpi_op_gemm", ptr(As[i % 2]), K, ptr(Ws[i % 6]), K, M, N, K, 1.0, ptr(bias), None, ptr(O), N, 1 << 30, None, 0, 0, 1, cfg, 0)
            rec[lab] = round(bench(go), 1)
        ctx.lib.pnpi_set_tuning(b"igemm_vpp", 0)
        print(json.dumps(rec), flush=True); out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/pp_ablate.json", "w"), indent=1)

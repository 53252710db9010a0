"""One-row GEMM shapes with the WEIGHT cold (rotating through > 400 MB of copies: every launch streams it from HBM, as inside a
forward whose 1.7 GB of weights never stay in the 256 MB Infinity Cache) against warm (one copy).  usage: wcold_probe.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"
def bench(go, n):
    for i in range(n): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K) in [(256, 1280, 1280), (1024, 640, 640), (4096, 320, 320), (64, 1280, 1280), (256, 10240, 1280), (256, 1280, 5120), (1024, 640, 2560), (4096, 320, 1280)]:
    wb = N * K * 2
    ncold = max(2, int(420e6 / wb) + 1)
    Ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).half() for _ in range(ncold)]
    bias = torch.randn(N, device=DEV)
    A = torch.randn(M, K, device=DEV).half(); O = torch.empty(M, N, device=DEV, dtype=torch.half)
    def mk(nc):
        def go(i):
            ctx.call("pnpi_op_gemm", ptr(A), K, ptr(Ws[i % nc]), K, M, N, K, 1.0, ptr(bias), None, ptr(O), N, 1 << 30, None, 0, 0, 1, -1, 0)
        return go
    n = max(3 * ncold, 30)
    warm, cold = bench(mk(1), n), bench(mk(ncold), n)
    print("(%d,%d,%d) W %.1f MB  warm %.1f us  cold %.1f us  (cold stream %.2f TB/s)" % (M, N, K, wb / 1e6, warm, cold, wb / cold / 1e6), flush=True)

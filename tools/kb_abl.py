import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); B = 4; DEV = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e-3
for (cin, hw, cout) in ((320, 64, 320), (640, 64, 320), (1280, 32, 640)):
    x = torch.randn(B, hw, hw, cin, device=DEV).half(); w = (torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half()
    bias = torch.randn(cout, device=DEV); out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
    t = timeit(lambda: ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), 0, 0))
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    print("conv %d@%d->%d cfg0: %.1f us  %.1f TF" % (cin, hw, cout, t * 1e6, fl / t / 1e12), flush=True)
M, K, N = 16384, 1280, 1280
a = torch.randn(M, K, device=DEV).half(); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).half(); o = torch.empty(M, N, device=DEV, dtype=torch.half)
t = timeit(lambda: ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, None, None, ptr(o), N, 1 << 30, None, 0, 0, 1, 0, 0))
print("gemm %dx%dx%d cfg0: %.1f us %.1f TF" % (M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))

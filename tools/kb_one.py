"""One conv + one GEMM shape, few launches -- for rocprofv3 PMC passes."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); B = 4; DEV = "cuda"
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for (cin, hw, cout) in ((320, 64, 320), (1280, 16, 1280)):
    x = torch.randn(B, hw, hw, cin, device=DEV).half(); w = (torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half()
    bias = torch.randn(cout, device=DEV); out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
    for _ in range(5):
        ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), cfg, 0)
M, K, N = 16384, 320, 2560
a = torch.randn(M, K, device=DEV).half(); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).half(); o = torch.empty(M, N, device=DEV, dtype=torch.half)
for _ in range(5):
    ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, None, None, ptr(o), N, 1 << 30, None, 0, 0, 1, cfg, 0)
torch.cuda.synchronize()

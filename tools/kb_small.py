"""Small-M (inversion pass, B=1) layer shapes with COLD weights (rotating through > 256 MB of distinct weight copies so neither
L2 nor the Infinity Cache holds them, as in a real forward).  usage: kb_small.py [force_cfg] [force_split]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(os.environ.get("B", "1"))
def run_conv(cin, hw, cout, label):
    ncopy = max(2, int(300e6 / (cout * 9 * cin * 2)) + 1)
    x = torch.randn(B, hw, hw, cin, device=DEV).half()
    ws = [(torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half() for _ in range(ncopy)]
    bias = torch.randn(cout, device=DEV); out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
    def go(i):
        ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(ws[i % ncopy]), ptr(bias), None, cout, ptr(out), cfg, split)
    for i in range(ncopy): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4 * ncopy
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    print("%-34s %7.1f us  %6.1f TF  W %.2f TB/s" % (label, us, fl / us / 1e6, cout * 9 * cin * 2 / us / 1e6))
def run_gemm(M, N, K, label):
    ncopy = max(2, int(300e6 / (N * K * 2)) + 1)
    a = torch.randn(M, K, device=DEV).half(); o = torch.empty(M, N, device=DEV, dtype=torch.half)
    ws = [(torch.randn(N, K, device=DEV) / math.sqrt(K)).half() for _ in range(ncopy)]
    def go(i):
        ctx.call("pnpi_op_gemm", ptr(a), K, ptr(ws[i % ncopy]), K, M, N, K, 1.0, None, None, ptr(o), N, 1 << 30, None, 0, 0, 1, cfg, split)
    for i in range(ncopy): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4 * ncopy
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("%-34s %7.1f us  %6.1f TF  W %.2f TB/s" % (label, us, 2.0 * M * N * K / us / 1e6, N * K * 2 / us / 1e6))
run_conv(1280, 8, 1280, "conv 8x8 1280->1280")
run_conv(2560, 8, 1280, "conv 8x8 2560->1280")
run_conv(1280, 16, 1280, "conv 16x16 1280->1280")
run_conv(640, 32, 640, "conv 32x32 640->640")
run_conv(320, 64, 320, "conv 64x64 320->320")
run_gemm(256 * B, 1280, 1280, "gemm M=256B N=1280 K=1280")
run_gemm(256 * B, 10240, 1280, "gemm M=256B N=10240 K=1280")
run_gemm(256 * B, 1280, 5120, "gemm M=256B N=1280 K=5120")
run_gemm(1024 * B, 640, 640, "gemm M=1024B N=640 K=640")
run_gemm(4096 * B, 320, 320, "gemm M=4096B N=320 K=320")
run_gemm(77 * B, 2560, 768, "gemm M=77B N=2560 K=768")

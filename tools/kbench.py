"""Kernel micro-benchmarks at the SD-1.x layer shapes (B = 4 UNet rows).  Prints TFLOP/s per kernel and writes
gpurun_out/kbench.json.  Usage: python tools/kbench.py [--quick]"""
import ctypes as C
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    quick = "--quick" in sys.argv
    ctx = Ctx()
    res = []
    B = 4
    convs = [(320, 64, 320), (640, 32, 640), (1280, 16, 1280), (1280, 8, 1280), (2560, 8, 1280), (960, 64, 320), (1920, 32, 640),
             (2560, 16, 1280)]
    for (cin, hw, cout) in convs:
        x = torch.randn(B, hw, hw, cin, device=DEV).half()
        w = (torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half()
        bias = torch.randn(cout, device=DEV)
        out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
        fl = 2.0 * B * hw * hw * cout * 9 * cin
        for cfg, split in ((-1, 0), (0, 0), (1, 0)):
            def f():
                ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), cfg, split)
            t = timeit(f, iters=5 if quick else 20)
            res.append({"op": "conv3x3", "cin": cin, "hw": hw, "cout": cout, "cfg": cfg, "ms": t * 1e3, "tflops": fl / t / 1e12})
            print(res[-1], flush=True)
    gemms = [(16384, 320, 2560), (16384, 1280, 320), (16384, 320, 1536), (4096, 640, 5120), (4096, 2560, 640), (1024, 1280, 10240),
             (1024, 5120, 1280), (256, 1280, 10240), (256, 5120, 1280), (16384, 320, 320)]
    for (M, K, N) in gemms:
        a = torch.randn(M, K, device=DEV).half()
        w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).half()
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV, dtype=torch.half)
        fl = 2.0 * M * N * K
        for cfg in (-1, 0, 1):
            def f():
                ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, ptr(bias), None, ptr(out), N, 1 << 30, None, 0, 0, 1, cfg, 0)
            t = timeit(f, iters=5 if quick else 20)
            res.append({"op": "gemm", "M": M, "K": K, "N": N, "cfg": cfg, "ms": t * 1e3, "tflops": fl / t / 1e12})
            print(res[-1], flush=True)
    heads = 8
    for (N, dh, Dp) in ((4096, 40, 64), (1024, 80, 96), (256, 160, 160)):
        hd = heads * Dp
        q = torch.randn(B, N, 2 * hd, device=DEV).half()
        vt = torch.randn(B, hd, N, device=DEV).half()
        o = torch.empty(B, N, heads * dh, device=DEV, dtype=torch.half)
        rows = torch.arange(B, dtype=torch.int32, device=DEV).repeat_interleave(4).reshape(B, 4).contiguous()
        fl = 4.0 * B * heads * N * N * dh

        def f():
            ctx.call("pnpi_op_attention", ptr(q), 2 * hd, 0, ptr(q), 2 * hd, hd, ptr(vt), N, ptr(o), heads * dh, heads, N, N, Dp, dh,
                     dh ** -0.5, ptr(rows), B)
        t = timeit(f, iters=5 if quick else 20)
        res.append({"op": "attn_self", "N": N, "dh": dh, "ms": t * 1e3, "tflops_alg": fl / t / 1e12})
        print(res[-1], flush=True)
    for (C, hw) in ((320, 64), (1280, 16), (2560, 8)):
        x = torch.randn(B, hw * hw, C, device=DEV).half()
        g = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
        out = torch.empty_like(x)

        def f():
            ctx.call("pnpi_op_groupnorm", ptr(x), None, C, 0, B, hw * hw, 32, 1e-5, ptr(g), ptr(b), 1, ptr(out))
        t = timeit(f, iters=5 if quick else 20)
        res.append({"op": "groupnorm", "C": C, "hw": hw, "ms": t * 1e3, "GBps": 3 * x.numel() * 2 / t / 1e9})
        print(res[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()

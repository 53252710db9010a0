"""the guide's 256^2 8-phase GEMM template (tools/micro/gemm_8phase.hip, written from cdna_hip_programming.md's description)
against igemm_pp<256,256> (tile id 16 through pnpi_op_gemm) and igemm_pp<192,320> (17), one process, interleaved arms, uniform random fp16
operands in [-1, 1) (and zero-filled operands for the clock effect), 4096^3 and 8192^3 (+ the long-K convolution panel shape).
usage: python tools/gemm_8phase_ab.py  ->  gpurun_out/gemm_8phase_ab.json"""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.gpu_util import Ctx, ptr  # noqa: E402

SRC = os.path.join(ROOT, "tools", "micro", "gemm_8phase.hip")
SO = os.path.join(ROOT, "tools", "micro", "libgemm8.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, SRC])
g8 = C.CDLL(SO)
g8.gemm8_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
ctx = Ctx()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


res = []
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (12288, 1280, 11520)]:
    for fill in ("random", "zero"):
        if fill == "random":
            a = (torch.rand(M, K, device="cuda") * 2 - 1).half()
            b = (torch.rand(N, K, device="cuda") * 2 - 1).half()
        else:
            a = torch.zeros(M, K, device="cuda", dtype=torch.half)
            b = torch.zeros(N, K, device="cuda", dtype=torch.half)
        c_ref = torch.empty(M, N, device="cuda", dtype=torch.half)
        c8 = torch.empty(M, N, device="cuda", dtype=torch.half)
        arms = {}
        for cfg in (16, 17):
            arms["igemm_pp_cfg%d" % cfg] = (lambda cfg=cfg: ctx.call("pnpi_op_gemm", ptr(a), K, ptr(b), K, M, N, K, 1.0, None, None, ptr(c_ref), N, 1 << 30, None, 0, 0, 1, cfg, 0))
        for var in (0, 1):
            def f(var=var):
                st = g8.gemm8_launch(ptr(a), ptr(b), ptr(c8), M, N, K, var, stream)
                assert st == 0, st
            arms["template_var%d" % var] = f
        for var, name in ((2, "ablation_no_A_dma"), (4, "ablation_no_B_dma"), (6, "ablation_no_dma")):
            def fa(var=var):
                st = g8.gemm8_launch(ptr(a), ptr(b), ptr(c8), M, N, K, var, stream)
                assert st == 0, st
            arms[name] = fa
        # correctness of the template against the product kernel (same accumulation order per element up to the k-step grouping)
        arms["igemm_pp_cfg16"]()
        chk = {}
        for var in (0, 1):
            c8.zero_()
            arms["template_var%d" % var]()
            torch.cuda.synchronize()
            d = (c8.float() - c_ref.float()).abs().max().item()
            chk["var%d" % var] = d / max(1e-9, c_ref.float().abs().max().item())
        iters = 20 if M * N * K < 2 ** 38 else 8
        times = {k: [] for k in arms}
        for rnd in range(4):                      # interleaved rounds
            for k, fn in arms.items():
                times[k].append(timeit(fn, iters))
        fl = 2.0 * M * N * K
        row = {"M": M, "N": N, "K": K, "fill": fill, "max_rel_diff_vs_cfg16": chk,
               "us": {k: [round(t * 1e6, 1) for t in v] for k, v in times.items()},
               "tflops_best": {k: round(fl / min(v) / 1e12, 1) for k, v in times.items()},
               "tflops_median": {k: round(fl / sorted(v)[len(v) // 2] / 1e12, 1) for k, v in times.items()}}
        res.append(row)
        print(json.dumps(row), flush=True)
        del a, b, c_ref, c8
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_8phase_ab.json"), "w"), indent=1)

"""Where the 8-phase template's advantage over igemm_pp<256,256> sits: per-K-tile slope or per-tile intercept (prologue + epilogue).
One tile per CU (M = 12288, N = 1280: 240 tiles), K = 2880 ... 23040, uniform random operands, interleaved arms.
usage: python tools/gemm_8phase_slope.py -> gpurun_out/gemm_8phase_slope.json"""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.gpu_util import Ctx, ptr  # noqa: E402
g8 = C.CDLL(os.path.join(ROOT, "tools", "micro", "libgemm8.so"))
g8.gemm8_launch.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
ctx = Ctx()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


M, N = 12288, 1280
Ks = [2880, 5760, 11520, 17280, 23040]
res = {"M": M, "N": N, "K": Ks, "us": {}}
for K in Ks:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half()
    b = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.half)
    arms = {"igemm_pp_cfg16": lambda: ctx.call("pnpi_op_gemm", ptr(a), K, ptr(b), K, M, N, K, 1.0, None, None, ptr(c), N, 1 << 30, None, 0, 0, 1, 16, 0),
            "template": lambda: g8.gemm8_launch(ptr(a), ptr(b), ptr(c), M, N, K, 0, stream),
            "template_no_dma": lambda: g8.gemm8_launch(ptr(a), ptr(b), ptr(c), M, N, K, 6, stream)}
    t = {k: [] for k in arms}
    for rnd in range(5):
        for k, fn in arms.items():
            t[k].append(timeit(fn))
    for k in arms:
        res["us"].setdefault(k, []).append(float(np.median(t[k])))
    print(K, {k: round(float(np.median(v)), 1) for k, v in t.items()}, flush=True)
for k, us in res["us"].items():
    nt = np.array(Ks) / 64.0
    slope, icpt = np.polyfit(nt, np.array(us), 1)
    res.setdefault("fit", {})[k] = {"us_per_ktile": float(slope), "intercept_us": float(icpt)}
print(json.dumps(res["fit"]))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_8phase_slope.json"), "w"), indent=1)

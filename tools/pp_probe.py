"""The 8-wave ping-pong GEMM kernel (cfg 16 = 256x256, 17 = 192x320) against the 4-wave tile configurations on the SD-1.x layer shapes of a
12-row launch (isolated launches, operands warm in the Infinity Cache: a first look -- the in-forward ranking comes from tools/fwd_tune.py).
Also the kernel's two ablations (tuning igemm_vpp: 1 = no MFMAs, 2 = no DMA).  usage: python tools/pp_probe.py -> gpurun_out/pp_probe.json"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr  # noqa: E402
DEV = "cuda"
ctx = Ctx()
lib = ctx.lib

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

res = []
B = 12
convs = [(320, 64, 320), (640, 32, 640), (1280, 16, 1280), (1280, 8, 1280), (960, 64, 320), (640, 64, 320), (1920, 32, 640), (2560, 16, 1280)]
CFGS = {(320, 64, 320): [(-1, 0), (14, 0), (4, 0), (16, 0), (17, 0)],
        (960, 64, 320): [(-1, 0), (4, 0), (17, 0)], (640, 64, 320): [(-1, 0), (4, 0), (17, 0)],
        (640, 32, 640): [(-1, 0), (14, 0), (16, 0), (17, 0), (17, 2), (16, 2)], (1920, 32, 640): [(-1, 0), (17, 2), (17, 3)],
        (1280, 16, 1280): [(-1, 0), (14, 2), (16, 4), (17, 4), (17, 3), (17, 5)], (2560, 16, 1280): [(-1, 0), (17, 4), (17, 8)],
        (1280, 8, 1280): [(-1, 0), (14, 8), (17, 16), (17, 12), (16, 16)]}
for (cin, hw, cout) in convs:
    x = torch.randn(B, hw, hw, cin, device=DEV).half()
    w = (torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half()
    bias = torch.randn(cout, device=DEV)
    out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    for cfg, split in CFGS[(cin, hw, cout)]:
        f = lambda: ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), cfg, split)
        t = timeit(f)
        res.append({"op": "conv3x3", "cin": cin, "hw": hw, "cout": cout, "cfg": cfg, "split": split, "us": t * 1e6, "tflops": fl / t / 1e12})
        print(res[-1], flush=True)
# ablations on the biggest conv shape
cin, hw, cout = 320, 64, 320
x = torch.randn(B, hw, hw, cin, device=DEV).half(); w = (torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half()
bias = torch.randn(cout, device=DEV); out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
for vpp in (0, 1, 2, 3, 4):
    assert lib.pnpi_set_tuning(b"igemm_vpp", vpp) == 0
    for cfg in (16, 17):
        f = lambda: ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), cfg, 0)
        t = timeit(f)
        res.append({"op": "ablation", "vpp": vpp, "cfg": cfg, "us": t * 1e6, "tflops_equiv": 2.0 * B * hw * hw * cout * 9 * cin / t / 1e12})
        print(res[-1], flush=True)
lib.pnpi_set_tuning(b"igemm_vpp", 0)
# channel-slab-major k-walk of the 3 x 3 convolutions (tuning igemm_tapin) against the default tap-major one, in turn
for (cin, hw, cout, cfg, split) in [(320, 64, 320, 17, 0), (960, 64, 320, 17, 0), (640, 32, 640, 17, 2), (1280, 16, 1280, 17, 4), (1280, 16, 1280, 16, 4), (2560, 16, 1280, 17, 4)]:
    x = torch.randn(B, hw, hw, cin, device=DEV).half(); w = (torch.randn(cout, 9 * cin, device=DEV) / math.sqrt(9 * cin)).half()
    bias = torch.randn(cout, device=DEV); out = torch.empty(B, hw, hw, cout, device=DEV, dtype=torch.half)
    f = lambda: ctx.call("pnpi_op_conv", ptr(x), None, cin, 0, B, hw, hw, 3, 1, 1, 0, hw, hw, ptr(w), ptr(bias), None, cout, ptr(out), cfg, split)
    for rnd in range(2):
        for tapin in (0, 1):
            assert lib.pnpi_set_tuning(b"igemm_tapin", tapin) == 0
            t = timeit(f)
            res.append({"op": "tapin", "cin": cin, "hw": hw, "cout": cout, "cfg": cfg, "split": split, "tapin": tapin, "us": t * 1e6, "tflops": 2.0 * B * hw * hw * cout * 9 * cin / t / 1e12})
            print(res[-1], flush=True)
lib.pnpi_set_tuning(b"igemm_tapin", 0)
gemms = [(49152, 320, 2560, [(-1, 0), (13, 0), (16, 0), (17, 0)]), (49152, 320, 1536, [(-1, 0), (0, 0), (16, 0)]), (49152, 1280, 320, [(-1, 0), (14, 0), (17, 0)]),
         (49152, 320, 320, [(-1, 0), (0, 0), (17, 0)]), (12288, 640, 5120, [(-1, 0), (0, 0), (16, 0), (17, 0)]), (12288, 2560, 640, [(-1, 0), (14, 0), (17, 0), (17, 2)]),
         (12288, 640, 640, [(-1, 0), (14, 0), (17, 0)]), (3072, 1280, 10240, [(-1, 0), (14, 0), (16, 0), (17, 0)]), (3072, 5120, 1280, [(-1, 0), (17, 4), (17, 2)]),
         (3072, 1280, 1280, [(-1, 0), (10, 0), (17, 0), (17, 2), (17, 4)]), (8192, 8192, 8192, [(0, 0), (14, 0), (16, 0), (17, 0)]), (4096, 4096, 4096, [(0, 0), (14, 0), (16, 0)])]
for (M, K, N, cfgs) in gemms:
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).half()
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.half)
    fl = 2.0 * M * N * K
    for cfg, split in cfgs:
        f = lambda: ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, ptr(bias), None, ptr(out), N, 1 << 30, None, 0, 0, 1, cfg, split)
        t = timeit(f)
        res.append({"op": "gemm", "M": M, "K": K, "N": N, "cfg": cfg, "split": split, "us": t * 1e6, "tflops": fl / t / 1e12})
        print(res[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/pp_probe.json", "w"), indent=1)

"""Where do the short-K (K = 320 ... 1280) 1x1 / linear GEMMs lose their time inside a forward?  One shape at a time, operands
either warm (one buffer, stays in the 256 MB Infinity Cache) or cold (rotating through > 300 MB of copies, as inside a forward
whose intermediate tensors flush the cache), under the ablation variants of the 128 x 128 kernel (DMA only / MFMA only), the
tile orders and the other tile configurations.  usage: shortk_probe.py -> stdout + gpurun_out/shortk_probe.json"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
ctx = Ctx(); DEV = "cuda"; lib = ctx.lib
def setk(**kw):
    for k, v in kw.items(): assert lib.pnpi_set_tuning(k.encode(), v) == 0, k
BASE = dict(igemm_force_cfg=-1, igemm_force_split=0, igemm_v128=2, tile_order=-1, igemm_table=1, igemm_bias_init=1)
def bench(go, ncopy):
    for i in range(ncopy): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(3 * ncopy, 12)
    e0.record()
    for i in range(n): go(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
out = {}
def probe(M, N, K, geglu=False, res=False):
    name = "(%d,%d,%d)%s%s" % (M, N, K, " geglu" if geglu else "", " +res" if res else "")
    No = N // 2 if geglu else N
    per_copy = M * K * 2 + M * No * 2 * (2 if res else 1)
    ncold = max(2, int(400e6 / per_copy) + 1)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).half(); bias = torch.randn(N, device=DEV)
    As = [torch.randn(M, K, device=DEV).half() for _ in range(ncold)]
    Os = [torch.empty(M, No, device=DEV, dtype=torch.half) for _ in range(ncold)]
    Rs = [torch.randn(M, No, device=DEV).half() for _ in range(ncold)] if res else None
    def mk(nc):
        def go(i):
            j = i % nc
            if geglu:
                ctx.call("pnpi_op_gemm_geglu", ptr(As[j]), K, ptr(w), K, M, N, K, ptr(bias), ptr(Os[j]), No)
            else:
                ctx.call("pnpi_op_gemm", ptr(As[j]), K, ptr(w), K, M, N, K, 1.0, ptr(bias), ptr(Rs[j]) if res else None, ptr(Os[j]), N, 1 << 30, None, 0, 0, 1, -1, 0)
        return go
    r = {}
    fl = 2.0 * M * N * K
    def rec(label, nc, **kn):
        setk(**{**BASE, **kn})
        us = bench(mk(nc), nc); r[label] = us
        print("%-28s %-26s %8.1f us %6.0f TF  io %.2f TB/s" % (name, label, us, fl / us / 1e6, per_copy / us / 1e6), flush=True)
    rec("table warm", 1)
    rec("table cold", ncold)
    rec("128 warm", 1, igemm_force_cfg=0)
    rec("128 cold", ncold, igemm_force_cfg=0)
    rec("128 cold bias in epilogue", ncold, igemm_force_cfg=0, igemm_bias_init=0)
    rec("128 cold dma-only", ncold, igemm_force_cfg=0, igemm_v128=11)
    rec("128 cold mfma-only", ncold, igemm_force_cfg=0, igemm_v128=12)
    rec("128 cold order1", ncold, igemm_force_cfg=0, tile_order=1)
    rec("128 cold order2", ncold, igemm_force_cfg=0, tile_order=2)
    rec("128 cold nst2 (4/CU)", ncold, igemm_force_cfg=13)
    rec("128 cold bk64x2", ncold, igemm_force_cfg=0, igemm_v128=0)
    rec("64 cold", ncold, igemm_force_cfg=1)
    rec("128x256 cold", ncold, igemm_force_cfg=5)
    rec("128x320 cold", ncold, igemm_force_cfg=4)
    rec("256x256 cold", ncold, igemm_force_cfg=7)
    rec("256x320 cold", ncold, igemm_force_cfg=6)
    rec("64x320 cold", ncold, igemm_force_cfg=12)
    out[name] = r
    setk(**BASE)
    del As, Os, Rs
    torch.cuda.empty_cache()
probe(49152, 2560, 320, geglu=True)
probe(49152, 2560, 320)
probe(49152, 320, 320, res=True)
probe(12288, 5120, 640, geglu=True)
probe(12288, 640, 640, res=True)
probe(3072, 1280, 1280, res=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/shortk_probe.json", "w"), indent=1)

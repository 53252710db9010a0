"""Quick per-kernel-class timing of full-width SD-1.x UNet forwards (B=1 and B=4) -- seconds, for kernel iteration."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine

def main():
    cfg = SD1
    ROWS = [int(x) for x in os.environ.get("ROWS", "1,4").split(",")]
    eng = NativeEngine(cfg, max_unet_rows=max(ROWS), max_vae_images=1)
    # random weights directly on the GPU (values irrelevant for timing, but not all-zero: DVFS)
    g = torch.Generator(device="cuda").manual_seed(0)
    usd = {k: (torch.randn(v.shape, device="cuda", generator=g) * 0.02) for k, v in weights.unet_state_dict.__wrapped__(cfg).items()} if hasattr(weights.unet_state_dict, "__wrapped__") else None
    if usd is None:
        import numpy as np
        # shapes only: build once on CPU with the cheap path
        sd = weights.unet_state_dict(cfg, 0)
        usd = {k: v.cuda() for k, v in sd.items()}
        vsd = {k: v.cuda() for k, v in weights.vae_state_dict(cfg, 0).items()}
    eng.load_state_dict(usd, vsd)
    res = {}
    for rows in ROWS:
        lat = torch.randn(rows, 4, 64, 64, device="cuda")
        ctx = torch.randn(rows, 77, 768, device="cuda")
        for _ in range(3):
            eng.unet(lat, 500, ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            eng.unet(lat, 500, ctx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        eng.profile_begin()
        for _ in range(3):
            eng.unet(lat, 500, ctx)
        st = eng.profile_end()
        print("rows=%d  %.2f ms/forward  %.1f TFLOP/s algorithmic" % (rows, dt * 1e3, rows * 803.27e9 / dt / 1e12))
        for k, v in st.items():
            if v["launches"]:
                tf = v["flops"] / (v["total_ms"] * 1e-3) / 1e12 if v["flops"] else 0
                gb = v["bytes"] / (v["total_ms"] * 1e-3) / 1e9 if v["bytes"] else 0
                print("   %-16s launches/fwd %5d  ms/fwd %7.3f  avg_us %7.2f  %7.1f TF  %7.1f GB/s" % (k, v["launches"] // 3, v["total_ms"] / 3, v["total_ms"] * 1e3 / v["launches"], tf, gb))
        res[rows] = {"ms": dt * 1e3, "classes": st}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/profile_forward.json", "w"))

if __name__ == "__main__":
    main()

"""Host enqueue time vs GPU time of one UNet forward (is the B=1 pass launch-bound on the CPU side?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
eng = NativeEngine(SD1, max_unet_rows=12, max_vae_images=1)
eng.load_state_dict({k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}, {k: v.cuda() for k, v in weights.vae_state_dict(SD1, 0).items()})
for rows in (1, 12):
    lat = torch.randn(rows, 4, 64, 64, device="cuda"); ctx = torch.randn(rows, 77, 768, device="cuda")
    for _ in range(3): eng.unet(lat, 500, ctx)
    torch.cuda.synchronize()
    n = 10; th = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); eng.unet(lat, 500, ctx); th += time.perf_counter() - a
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("rows=%d host enqueue %.2f ms/forward, wall %.2f ms/forward" % (rows, th / n * 1e3, tt / n * 1e3))

"""N UNet forwards at a given row count and nothing else (for rocprofv3 kernel-trace: kernel time vs wall time)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
rows = int(os.environ.get("ROWS", "1")); n = int(os.environ.get("N", "20"))
eng = NativeEngine(SD1, max_unet_rows=max(rows, 4), max_vae_images=1)
eng.load_state_dict({k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}, {k: v.cuda() for k, v in weights.vae_state_dict(SD1, 0).items()})
for kv in filter(None, os.environ.get("TUNE", "").split(",")):      # TUNE="igemm_deep_rings=0,igemm_table=0"
    k, v = kv.split("="); assert eng.lib.pnpi_set_tuning(k.encode(), int(v)) == 0, k
lat = torch.randn(rows, 4, 64, 64, device="cuda"); ctx = torch.randn(rows, 77, 768, device="cuda")
for _ in range(int(os.environ.get("WARM", "3"))): eng.unet(lat, 500, ctx)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): eng.unet(lat, 500, ctx)
torch.cuda.synchronize(); print("rows=%d wall %.3f ms/forward over %d" % (rows, (time.perf_counter() - t0) / n * 1e3, n))
if os.environ.get("DUMP"):          # per-launch records (class, shape, HIP-event us, algorithmic flops / bytes, kernel template) of 3 more forwards
    os.environ["PNPI_PROFILE_DUMP"] = os.environ["DUMP"]
    eng.profile_begin()
    for _ in range(3): eng.unet(lat, 500, ctx)
    eng.profile_end()

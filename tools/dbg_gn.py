import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
eng = NativeEngine(SD1, max_unet_rows=4, max_vae_images=1)
g = torch.Generator(device="cuda").manual_seed(0)
usd = {k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}
eng.load_state_dict(usd, None)
eng.mark_all_loaded()
lat = torch.randn(4, 4, 64, 64, device="cuda"); ctx = torch.randn(4, 77, 768, device="cuda")
os.environ["PNPI_GN_DEBUG"] = "1"
eng.unet(lat, 500, ctx); torch.cuda.synchronize()

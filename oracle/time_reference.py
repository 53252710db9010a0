"""Test infrastructure (build container only: /root/reference is not on the GPU box).  Times the reference's OWN UNet module
(models/edict/my_diffusers UNet2DConditionModel through oracle/ref_shim.py, full SD-1.x width, fp32, this container's CPU threads)
next to the oracle's restatement of it on the same weights and inputs -- the check that bench.py's cpu_baseline (kind "port": the
oracle timed on the GPU box's host) stands for the reference's CPU speed.  usage: python oracle/time_reference.py [rows]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, sd_oracle
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
assert ref_shim.available(), "/root/reference not present"
ref_shim.install()
cfg = SD1
sd = weights.unet_state_dict(cfg, 0)
unet = ref_shim.build_unet(cfg, sd)
g = torch.Generator().manual_seed(3)
lat = torch.randn(rows, 4, 64, 64, generator=g)
ctx = weights.synth_context(cfg, rows, seed=4).cpu()
t_ref = t_or = 1e30
with torch.no_grad():
    for _ in range(3):            # interleaved, best of 3 (the first pass pays page faults and thread-pool start-up)
        t0 = time.perf_counter(); a = unet(lat, 500, encoder_hidden_states=ctx)["sample"]; t_ref = min(t_ref, time.perf_counter() - t0)
        t0 = time.perf_counter(); b = sd_oracle.unet_forward(sd, cfg, lat, 500, ctx); t_or = min(t_or, time.perf_counter() - t0)
rel = ((a - b).norm() / a.norm()).item()
print("threads %d rows %d: reference UNet %.2f s, oracle %.2f s (ratio %.2f), rel-L2 %.1e" % (torch.get_num_threads(), rows, t_ref, t_or, t_or / t_ref, rel))

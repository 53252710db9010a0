"""Checker script (GPU box; lives under tests/ because it uses the oracle): per-site attention probabilities of the materialise-and-call-back
path vs the CPU oracle's, all 32 sites.  usage: python tests/cb_sites_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import TINY16
from pnpinversion_amd.engine import NativeEngine
from oracle import sd_oracle
cfg = TINY16
usd, vsd = weights.unet_state_dict(cfg, 1), weights.vae_state_dict(cfg, 1)
eng = NativeEngine(cfg, max_unet_rows=8, max_vae_images=2); eng.load_state_dict(usd, vsd)
def rel(a, b): return ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm()).item()
g = torch.Generator().manual_seed(51)
lat = torch.randn(4, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g).cuda()
ctx = weights.synth_context(cfg, 4, seed=52)
ref_sites = []
def ohook(attn, is_cross, place):
    ref_sites.append(attn.detach().clone()); return attn
with torch.no_grad(): ref = sd_oracle.unet_forward(usd, cfg, lat.cpu(), 500, ctx.cpu(), ohook)
nat_sites = []
def nhook(attn, is_cross, place, layer): nat_sites.append(attn.detach().float().cpu().clone())
eng.set_attention_callback(nhook, rows=4)
cb = eng.unet(lat, 500, ctx).clone()
eng.set_attention_callback(None)
fused = eng.unet(lat, 500, ctx)
print("cb-vs-oracle %.2e fused-vs-oracle %.2e" % (rel(cb, ref), rel(fused, ref)))
print("heads", cfg.heads, "shapes", [tuple(s.shape) for s in nat_sites[:4]], [tuple(s.shape) for s in ref_sites[:4]])
for i, (a, b) in enumerate(zip(nat_sites, ref_sites)):
    b = b.reshape(a.shape)
    print("site %2d %s rel %.2e  rowsum min %.4f max %.4f  max|d| %.2e" % (i, tuple(a.shape), rel(a, b), a.sum(-1).min(), a.sum(-1).max(), (a - b).abs().max()))

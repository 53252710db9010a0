"""Which layer shape goes wrong under the ping-pong kernel: the full-width 2 + 2-step lock-step edit (tests/golden/e2e_sd1.npz) with the table's
ping-pong entries enabled for one (N, K) at a time (tuning igemm_pp_only_n / _k; igemm_pp_only_n = -1: none)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.p2p import attention_control as ac
from pnpinversion_amd.pipeline import NativePipeline
from pnpinversion_amd.text import SyntheticTextEncoder
g = np.load("tests/golden/e2e_sd1.npz")
cfg, steps = SD1, int(g["steps"])
pipe = NativePipeline(cfg, max_unet_rows=12, max_vae_images=2, text_encoder=SyntheticTextEncoder(cfg.cross_dim, seed=7))
pipe.load_state_dict(weights.unet_state_dict(cfg, 0), weights.vae_state_dict(cfg, 0))
eng = pipe.engine
lib = eng.lib
pipe.scheduler.set_timesteps(steps)
ts = pipe.scheduler.timesteps.numpy()
ctx = torch.from_numpy(g["context"]).float()
x_stars = torch.from_numpy(g["x_stars"])
w0, w1 = [str(x) for x in g["blend"]]
ctrl = ac.make_controller(pipe, [str(g["src"]), str(g["tgt"])], False, {"default_": 0.4}, 0.6, ((w0,), (w1,)), {"words": (w1,), "values": (2,)}, num_ddim_steps=steps)
def run(tag):
    nl, lats = eng.direct_edit(x_stars, ctx[None], [None, [ctrl.tables()]], ts, 7.5)
    torch.cuda.synchronize()
    e = lats[1, 0][1].cpu(); ref = torch.from_numpy(g["edited_latents"])[1]
    bad = int(torch.isnan(lats).sum())
    r = ((e - ref).norm() / ref.norm()).item()
    print("%-28s nan=%d  edit rel=%.3e  recon rel=%.3e" % (tag, bad, r, ((lats[0, 0][1].cpu() - torch.from_numpy(g["reconstruct_latent"])[1]).norm() / ref.norm()).item()), flush=True)
    return bad
def setk(**kw):
    for k, v in kw.items(): assert lib.pnpi_set_tuning(k.encode(), v) == 0, k
run("default"); run("default again")
setk(igemm_pp_only_n=-1); run("no pp"); setk(igemm_pp_only_n=0)
shapes = set()
for line in open("pnpinversion_amd/csrc/tile_table.inc"):
    if line.startswith("{"):
        M, N, K, ks, c, s = [int(x) for x in line[1:line.index("}")].split(",")]
        if c in (16, 17) and M in (49152, 12288, 3072, 768): shapes.add((M, N, K))
for (M, N, K) in sorted(shapes):
    setk(igemm_pp_only_m=M, igemm_pp_only_n=N, igemm_pp_only_k=K)
    run("pp only M=%d N=%d K=%d" % (M, N, K))
setk(igemm_pp_only_m=0, igemm_pp_only_n=0, igemm_pp_only_k=0)

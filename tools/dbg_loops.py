import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SMALL64
from pnpinversion_amd.pipeline import NativePipeline
from pnpinversion_amd.text import SyntheticTextEncoder
from pnpinversion_amd.p2p import attention_control as ac
def rel(a,b):
    a,b=torch.as_tensor(a).float().cpu(),torch.as_tensor(b).float().cpu(); return ((a-b).norm()/b.norm()).item()
cfg=SMALL64
pipe=NativePipeline(cfg,max_unet_rows=4,max_vae_images=2,text_encoder=SyntheticTextEncoder(cfg.cross_dim,seed=7))
pipe.load_state_dict(weights.unet_state_dict(cfg,2),weights.vae_state_dict(cfg,2))
eng=pipe.engine
for name in ("refine","replace"):
    g=np.load("tests/golden/e2e_%s.npz"%name); steps=int(g["steps"]); pipe.scheduler.set_timesteps(steps); ts=pipe.scheduler.timesteps.numpy()
    ctx=torch.from_numpy(g["context"]).float(); xs=torch.from_numpy(g["x_stars"]); ref_nl=torch.from_numpy(g["noise_loss"])
    got=eng.ddim_invert(xs[0],ctx[2:3],ts)
    print(name,"invert rel",[rel(got[i],xs[i]) for i in range(steps+1)])
    nl=eng.offset_calculate(xs,ctx[None],ts,7.5)
    print(" offsets rel",[rel(nl[i,0],ref_nl[i]) for i in range(steps)],"max abs",(nl[:,0].cpu()-ref_nl).abs().max().item(),"ref rms",ref_nl.pow(2).mean().sqrt().item(),"lat rms",xs[0].pow(2).mean().sqrt().item())
    rec=eng.edit_loop(xs[-1],ctx[None],ref_nl[:,None],None,ts,7.5)[0]
    print(" recon rel",rel(rec,g["reconstruct_latent"]), "rows", rel(rec[0],g["reconstruct_latent"][0]), rel(rec[1],g["reconstruct_latent"][1]))
    w0,w1=[str(x) for x in g["blend"]]; ub=bool(g["use_blend"])
    ctrl=ac.make_controller(pipe,[str(g["src"]),str(g["tgt"])],bool(g["is_replace"]),{"default_":0.4},0.6,((w0,),(w1,)) if ub else None,{"words":(w1,),"values":(2,)} if ub else None,num_ddim_steps=steps)
    out=eng.edit_loop(xs[-1],ctx[None],ref_nl[:,None],[ctrl.tables()],ts,7.5)[0]
    ref=torch.from_numpy(g["edited_latents"])
    d=(out.cpu()-ref).abs().amax(dim=-3)
    print(" edit rel",rel(out,ref),"row0",rel(out[0],ref[0]),"row1",rel(out[1],ref[1]),"frac>0.25",(d>0.25).float().mean().item(),"frac>0.1",(d>0.1).float().mean().item(), "max", d.max().item())

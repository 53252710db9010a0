"""Full-width null-text optimisation timing: N DDIM steps x INNER Adam iterations at one UNet row (recording forward + reverse walk +
Adam), with per-launch-class HIP-event sums of one iteration block.  STEPS=2 INNER=10 by default."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
from pnpinversion_amd.p2p.scheduler_dev import DDIMSchedulerDev
steps = int(os.environ.get("STEPS", "2")); inner = int(os.environ.get("INNER", "10"))
eng = NativeEngine(SD1, max_unet_rows=12, max_vae_images=1)
eng.load_state_dict({k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}, {k: v.cuda() for k, v in weights.vae_state_dict(SD1, 0).items()})
sch = DDIMSchedulerDev(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
sch.bind(eng); sch.set_timesteps(steps)
g = torch.Generator().manual_seed(0)
z0 = torch.randn(1, 4, 64, 64, generator=g).cuda()
ctx = weights.synth_context(SD1, 2, seed=3).cuda()
xs = eng.ddim_invert(z0, ctx[1:], sch.timesteps.numpy())
torch.cuda.synchronize()
for rep in range(2):
    eng.reset_counters()
    t0 = time.perf_counter()
    emb, its, losses = eng.null_text_optimize(xs, ctx[:1], ctx[1:], sch.timesteps.numpy(), 7.5, num_inner_steps=inner, epsilon=1e-5, return_losses=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = eng.counters()
    n_it = sum(its)
    print("rep %d: %d steps, %d Adam iterations, %.3f s  -> %.1f ms per iteration (forward + backward + Adam); forwards %d backward rows %d"
          % (rep, steps, n_it, dt, dt / max(1, n_it) * 1e3, c["unet_sample_forwards"], c.get("unet_backward_rows", -1)))
print("losses step 0:", ["%.5f" % l for l in losses[0]])
print("free/total GB", [x / 2**30 for x in torch.cuda.mem_get_info()])
if os.environ.get("PROFILE"):
    os.environ["PNPI_PROFILE_DUMP"] = os.environ.get("DUMP", "gpurun_out/null_text_launches.csv")
    eng.profile_begin()
    eng.null_text_optimize(xs[-2:], ctx[:1], ctx[1:], sch.timesteps.numpy()[:1], 7.5, num_inner_steps=3, epsilon=1e-5)
    cls = eng.profile_end()
    print("classes over 1 step x 3 iterations (launches, ms):", {k: (v["launches"], round(v["total_ms"], 2)) for k, v in cls.items() if v["launches"]})

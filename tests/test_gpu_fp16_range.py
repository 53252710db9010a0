"""fp16 range with adversarially scaled weights ("what's weak" 2 / next 8).

The seeded synthetic weights keep every activation O(1); real SD-1.x checkpoints do not (residual-stream outliers of 1e3 and more,
attention logits of +-50).  Here the TINY16 weights are rescaled so that the device path sees that regime -- `conv_in` x 2000 (the
residual stream of every ResNet / transformer block carries |h| up to ~1e4, NHWC fp16 activations within 6x of the fp16 maximum),
every to_q / to_k x 4 (logits x 16: standard deviation ~16, extremes beyond +-50) -- and held against the fp32 CPU oracle on the SAME
(fp16-representable) parameters at the tolerances DESIGN.md section 4 states: one UNet forward, one 5 + 5-step direct-inversion edit loop, the
context gradient and a 3-step null-text optimisation (10 Adam iterations each, fp16 activation gradients under the fixed 2^12 loss scale).
Measured values go to gpurun_out/fp16_range.json."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import p2p_oracle as po  # noqa: E402   (checker only)
from oracle import sd_oracle  # noqa: E402
from pnpinversion_amd import weights  # noqa: E402
from pnpinversion_amd.config import TINY16  # noqa: E402
from pnpinversion_amd.p2p import attention_control as ac  # noqa: E402
from pnpinversion_amd.pipeline import NativePipeline  # noqa: E402
from pnpinversion_amd.text import SyntheticTextEncoder  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAM_SCALE, QK_SCALE = 2000.0, 4.0


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _log(key, vals):
    path = os.path.join(ROOT, "gpurun_out", "fp16_range.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = vals
    json.dump(d, open(path, "w"), indent=1)


def stressed_unet_state_dict(cfg, seed):
    """weights.unet_state_dict with the residual stream and the attention logits blown up; values stay fp16-representable."""
    sd = dict(weights.unet_state_dict(cfg, seed))
    for k in list(sd):
        if k.startswith("conv_in."):
            sd[k] = (sd[k].float() * STREAM_SCALE).half().float()
        elif k.endswith(".to_q.weight") or k.endswith(".to_k.weight"):
            sd[k] = (sd[k].float() * QK_SCALE).half().float()
    assert torch.isfinite(sd["conv_in.weight"]).all()
    return sd


@pytest.fixture(scope="module")
def stressed():
    cfg = TINY16
    usd = stressed_unet_state_dict(cfg, 5)
    pipe = NativePipeline(cfg, max_unet_rows=12, max_vae_images=1, text_encoder=SyntheticTextEncoder(cfg.cross_dim, seed=3))
    pipe.load_state_dict(usd, weights.vae_state_dict(cfg, 5))
    yield cfg, usd, pipe
    pipe.engine.close()


def test_the_stress_is_real(stressed):
    """conv_in's output -- what every residual connection of the first level carries -- reaches the 1e3 .. 1e4 range, and the first
    self-attention's logits the +-50 range (computed here in fp32 from the same parameters)."""
    cfg, usd, _ = stressed
    g = torch.Generator().manual_seed(41)
    lat = torch.randn(4, 4, cfg.sample_size, cfg.sample_size, generator=g)
    h = F.conv2d(lat, usd["conv_in.weight"], usd["conv_in.bias"], padding=1)
    amax = h.abs().max().item()
    # logits of down_blocks.0.attentions.0 attn1 on a unit-variance input (what LayerNorm hands it)
    C = cfg.block_out_channels[0]
    x = torch.randn(256, C, generator=g)
    pre = "down_blocks.0.attentions.0.transformer_blocks.0.attn1."
    q, k = x @ usd[pre + "to_q.weight"].t(), x @ usd[pre + "to_k.weight"].t()
    dh = C // cfg.heads
    logits = (q.view(256, cfg.heads, dh).transpose(0, 1) @ k.view(256, cfg.heads, dh).transpose(0, 1).transpose(1, 2)) * dh ** -0.5
    _log("stress", {"conv_in_abs_max": amax, "conv_in_rms": h.pow(2).mean().sqrt().item(), "logit_std": logits.std().item(),
                    "logit_abs_max": logits.abs().max().item()})
    assert 1e3 < amax < 6.5e4, amax
    assert logits.abs().max().item() > 50.0, logits.abs().max().item()


def test_unet_forward_under_stress(stressed):
    cfg, usd, pipe = stressed
    eng = pipe.engine
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(4, 4, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = weights.synth_context(cfg, 4, seed=43)
    with torch.no_grad():
        ref = sd_oracle.unet_forward(usd, cfg, lat, 500, ctx)
    got = eng.unet(lat.cuda(), 500, ctx.cuda()).cpu()
    r = rel(got, ref)
    _log("unet_forward", {"rel_l2": r, "ref_abs_max": ref.abs().max().item()})
    assert torch.isfinite(got).all()
    assert r < 4e-3, r                               # UNet forward <= 4e-3


def test_edit_loops_under_stress(stressed):
    """5 + 5 steps: DDIM inversion, offsets, Replace + Reweight edit pass -- the tolerances of test_loops_against_oracle_tiny."""
    cfg, usd, pipe = stressed
    eng = pipe.engine
    steps = 5
    pipe.scheduler.set_timesteps(steps)
    ts = pipe.scheduler.timesteps.numpy()
    g = torch.Generator().manual_seed(21)
    z0 = torch.randn(1, 4, 16, 16, generator=g)
    ctx = weights.synth_context(cfg, 4, seed=22)
    ac_ = po.alphas_cumprod()

    def unet_fn(lat, t, c, hook):
        with torch.no_grad():
            return sd_oracle.unet_forward(usd, cfg, lat, t, c, hook)

    ref_lat = po.ddim_loop(unet_fn, z0, ctx[2:3], po.make_timesteps(steps), ac_, ac_[0])
    got_lat = eng.ddim_invert(z0, ctx[2:3], ts)
    r_inv = rel(got_lat, torch.stack(ref_lat))
    ref_nl = po.offset_calculate(unet_fn, ref_lat, ctx, po.make_timesteps(steps), ac_, ac_[0], 7.5)
    got_nl = eng.offset_calculate(torch.stack(ref_lat), ctx[None], ts, 7.5)
    r_nl = rel(got_nl[:, 0], torch.stack(ref_nl))
    prompts = ["a round cake with orange frosting on a wooden plate", "a square cake with orange frosting on a wooden plate"]
    c = ac.make_controller(pipe, prompts, True, {"default_": 0.4}, 0.6, None, {"words": ("square",), "values": (2,)}, num_ddim_steps=steps)
    tables = {"kind": "replace", "mapper": c.prev_controller.mapper[0], "equalizer": c.equalizer.reshape(-1),
              "cross_alpha": c.cross_replace_alpha.reshape(steps + 1, 77), "self_range": c.num_self_replace, "lb": None}
    ref_out = po.guidance_forward(unet_fn, ref_lat[-1], ctx, ref_nl, po.EditController(32, tables), po.make_timesteps(steps), ac_, ac_[0], 7.5)
    got_out = eng.edit_loop(ref_lat[-1], ctx[None], got_nl, [c.tables()], ts, 7.5)[0]
    r_edit, r_src = rel(got_out[1], ref_out[1]), rel(got_out[0], z0[0])
    _log("loops", {"inversion": r_inv, "offsets": r_nl, "edited": r_edit, "source_row": r_src})
    assert torch.isfinite(got_lat).all() and torch.isfinite(got_nl).all() and torch.isfinite(got_out).all()
    assert r_inv < 4e-3 * steps ** 0.5, r_inv
    assert r_nl < 1.5e-2, r_nl
    assert r_edit < 2e-2 and r_src < 2e-2, (r_edit, r_src)


def test_context_gradient_and_null_text_step_under_stress(stressed):
    """The backward pass in the same regime: d eps / d context against autograd through the oracle, and three DDIM steps of the null-text
    optimisation (10 Adam iterations each) against oracle/p2p_oracle.null_optimization -- every iteration's loss and the embeddings."""
    cfg, usd, pipe = stressed
    eng = pipe.engine
    g = torch.Generator().manual_seed(31)
    lat = torch.randn(1, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = weights.synth_context(cfg, 1, seed=32).cpu()
    d_eps = torch.randn(1, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    cr = ctx.clone().requires_grad_(True)
    eps_ref = sd_oracle.unet_forward(usd, cfg, lat, 500, cr)
    eps_ref.backward(d_eps)
    out = {}
    for scale in (256.0, 4096.0):                       # the test's own scale and the optimisation loop's fixed 2^12
        eps, dctx = eng.unet_context_grad(lat.cuda(), 500, ctx.cuda(), (d_eps * scale).cuda())
        got = dctx.cpu() / scale
        out["grad_rel_scale_%d" % int(scale)] = rel(got, cr.grad)
        assert torch.isfinite(got).all()
    out["eps_rel"] = rel(eps.cpu(), eps_ref.detach())
    out["grad_norm"] = cr.grad.norm().item()
    # the optimisation loop (fixed 2^12 loss scale inside the library)
    steps = 3
    pipe.scheduler.set_timesteps(steps)
    ts = pipe.scheduler.timesteps.numpy()
    ac_ = po.alphas_cumprod()
    ctx2 = weights.synth_context(cfg, 2, seed=33).cpu()

    def unet_fn(l, t, c, hook):
        return sd_oracle.unet_forward(usd, cfg, l, t, c, hook)

    with torch.no_grad():
        ref_lat = po.ddim_loop(lambda l, t, c, h: sd_oracle.unet_forward(usd, cfg, l, t, c, h), lat, ctx2[1:2], po.make_timesteps(steps), ac_, ac_[0])
    trace = []
    ref_unc = po.null_optimization(unet_fn, ref_lat, ctx2[:1], ctx2[1:2], po.make_timesteps(steps), ac_, ac_[0], 7.5, num_inner_steps=10,
                                   epsilon=1e-5, trace=trace)
    got, its, losses = eng.null_text_optimize(torch.stack(ref_lat).cuda(), ctx2[:1].cuda(), ctx2[1:2].cuda(), ts, 7.5, num_inner_steps=10,
                                              epsilon=1e-5, return_losses=True)
    ref_losses = torch.tensor([l for tr in trace for l in tr[3]])
    got_losses = torch.tensor([l for ls in losses for l in ls])
    out["null_text_iterations"] = {"ref": [tr[1] for tr in trace], "got": list(its)}
    out["null_text_losses_ref"] = ref_losses.tolist()
    out["null_text_losses"] = got_losses.tolist()
    n = min(len(ref_losses), len(got_losses))
    out["null_text_loss_max_dev"] = (got_losses[:n] / ref_losses[:n] - 1).abs().max().item()
    out["null_text_embedding_rel"] = rel(got.cpu(), torch.stack(ref_unc))
    _log("backward", out)
    assert out["eps_rel"] < 4e-3
    # measured on MI355X: 3.35e-2 at BOTH loss scales (2^8 and the loop's 2^12: neither underflow nor overflow -- the error is the fp16
    # rounding of the activation gradients behind near-one-hot attention rows, logits up to +-180 here); unstressed the same check
    # measures 1.2e-2 .. 2.2e-3 (TINY16 .. full width) against a 3e-2 bar
    assert out["grad_rel_scale_256"] < 5e-2 and out["grad_rel_scale_4096"] < 5e-2, out
    assert abs(out["grad_rel_scale_256"] - out["grad_rel_scale_4096"]) < 2e-3, out          # the loss scale is not what limits it
    assert len(got_losses) == len(ref_losses), (len(got_losses), len(ref_losses))
    assert out["null_text_loss_max_dev"] < 2e-2, out["null_text_loss_max_dev"]
    # Adam's first updates are lr * sign(g): gradient elements within fp16 noise of zero land 2 * lr away (measured 1.8e-2 here, 6.5e-3
    # unstressed); the LOSSES -- what the optimisation is for -- track the oracle's to 7e-4
    assert out["null_text_embedding_rel"] < 2.5e-2, out["null_text_embedding_rel"]

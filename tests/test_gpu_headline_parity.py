"""Loop-level parity on the BENCHMARKED configuration and schedule ("what's weak" 1-2):

  (a) full SD-1.x width (the 859.5 M-parameter UNet + 83.7 M VAE bench.py times), the reference's own
      P2PEditor("directinversion+p2p") run at 2 + 2 steps with Refine + Reweight + LocalBlend (tests/golden/e2e_sd1.npz, made by
      oracle/make_golden.py from the unmodified reference modules): inversion latents, offsets, reconstruction / edited latents and
      the decoded panels through pnpi_ddim_invert / pnpi_direct_edit / P2PEditor;
  (a2) the same at the benchmarked SCHEDULE: full width, 50 + 50 steps, against the reference's own run (tests/golden/e2e_sd1_50.npz) --
      this is also the long-schedule check of the 64 x 64 (4096-token) attention sites, which the reduced 50-step configurations below
      leave out; measured on MI355X: inversion <= 2.0e-3 at every sampled step, offsets <= 1.7e-3, edited latents 4.9e-3, panels 58 / 52 dB;
  (b) the FULL 50 + 50-step schedule with its real tables (51-row cross_replace_alpha, LocalBlend from step 10, self-attention
      window 0..30) on the reduced-width configurations against the CPU oracle, asserting the tolerances SURVEY.md 8(d) states for
      50 + 50 steps -- final latents rel-L2 <= 2e-2, decoded images PSNR >= 35 dB (and mean |diff| <= 2/255) -- and logging the
      per-step drift of the inversion trajectory and of the offsets (gpurun_out/drift_<cfg>.json);
  (c) AttentionRefine with INSERTED target tokens (seq_aligner.get_mapper's -1 / alpha 0 entries) against the reference's own run
      (tests/golden/e2e_insert2.npz: with LocalBlend + Reweight, e2e_insert3.npz: plain Refine).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import p2p_oracle as po  # noqa: E402   (checker only)
from oracle import sd_oracle  # noqa: E402
from pnpinversion_amd import weights  # noqa: E402
from pnpinversion_amd.config import SD1, SMALL64, SMALL64_LB, TINY16  # noqa: E402
from pnpinversion_amd.p2p import attention_control as ac  # noqa: E402
from pnpinversion_amd.p2p_editor import P2PEditor  # noqa: E402
from pnpinversion_amd.pipeline import NativePipeline  # noqa: E402
from pnpinversion_amd.text import SyntheticTextEncoder  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


def masked_rel(a, b, pix_tol=0.25):
    """rel-L2 outside the latent pixels where a LocalBlend mask decision (a hard > 0.3 threshold) fell on the other side."""
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    bad = (a - b).abs().amax(dim=-3) > pix_tol
    keep = (~bad).unsqueeze(-3).expand_as(a)
    return ((a - b)[keep].norm() / b[keep].norm()).item(), bad.float().mean().item()


def psnr_u8(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def _cat_image():
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLD, "example_cat_512.png")))[:, :, :3]


# ----------------------------------------------------------------------------------------------- (a) full SD-1.x width
@pytest.mark.parametrize("fixture", ["e2e_sd1.npz", "e2e_sd1_replace.npz"])
def test_sd1_full_width_loops_and_editor_against_reference_golden(fixture):
    """e2e_sd1.npz: Refine + Reweight + LocalBlend (the PIE-Bench default).  e2e_sd1_replace.npz (round 6): BASELINE
    config 2's "P2P AttentionReplace" -- is_replace_controller=True (the 77 x 77 replacement mapper of attention_control.py:301-314 in
    attn_cross_edit_kernel at 320 / 640 / 1280 channels) + Reweight + LocalBlend on the cake pair, the reference's own run."""
    g = np.load(os.path.join(GOLD, fixture))
    is_replace = bool(g["is_replace"])
    cfg, steps = SD1, int(g["steps"])
    pipe = NativePipeline(cfg, max_unet_rows=12, max_vae_images=2, text_encoder=SyntheticTextEncoder(cfg.cross_dim, seed=7))
    seed = int(g["weight_seed"])
    pipe.load_state_dict(weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed))
    eng = pipe.engine
    pipe.scheduler.set_timesteps(steps)
    ts = pipe.scheduler.timesteps.numpy()
    ctx = torch.from_numpy(g["context"]).float()
    x_stars = torch.from_numpy(g["x_stars"])
    got = eng.ddim_invert(x_stars[0], ctx[2:3], ts)
    r_inv = rel(got, x_stars)
    assert r_inv < 4e-3 * steps ** 0.5, r_inv
    src, tgt = str(g["src"]), str(g["tgt"])
    w0, w1 = [str(x) for x in g["blend"]]
    ctrl = ac.make_controller(pipe, [src, tgt], is_replace, {"default_": 0.4}, 0.6, ((w0,), (w1,)), {"words": (w1,), "values": (2,)},
                              num_ddim_steps=steps)
    inner = ctrl.prev_controller if isinstance(ctrl, ac.AttentionReweight) else ctrl
    assert isinstance(inner, ac.AttentionReplace if is_replace else ac.AttentionRefine)
    # the product's schedule: offsets + reconstruction + edit pass, one 12-row launch per timestep, from the reference's trajectory
    nl, lats = eng.direct_edit(x_stars, ctx[None], [None, [ctrl.tables()]], ts, 7.5)
    r_nl = rel(nl[:, 0], g["noise_loss"])
    assert r_nl < 1.5e-2, r_nl
    r_rec = rel(lats[0, 0][1], g["reconstruct_latent"][1])
    assert r_rec < 1.5e-2, r_rec
    r_edit, frac = masked_rel(lats[1, 0][1], torch.from_numpy(g["edited_latents"])[1])
    assert frac <= 0.005 and r_edit < 1.5e-2, (r_edit, frac)
    assert rel(lats[1, 0][0], x_stars[0][0]) < 2e-2                      # the source branch reproduces x*_0 (SURVEY Note D)
    assert torch.equal(lats[0, 0][0], lats[1, 0][0])                     # same computation in the same launch: bit-identical
    # drop-in API end to end at full width (VAE at 512 x 512 included): panels of the reference's own run, 4x subsampled
    ed = P2PEditor(["directinversion+p2p"], "cuda", num_ddim_steps=steps, pipeline=pipe)
    panel, st = ed.edit_image_directinversion(_cat_image(), src, tgt, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
                                              blend_word=((w0,), (w1,)), eq_params={"words": (w1,), "values": (2,)}, return_stages=True,
                                              is_replace_controller=is_replace)
    xs = torch.stack([x.cpu() for x in st["x_stars"]])
    assert rel(xs, g["x_stars"]) < 6e-3, rel(xs, g["x_stars"])          # includes the full-width 512 x 512 VAE encode
    r_e2e, frac = masked_rel(st["latents"], torch.from_numpy(g["edited_latents"]))
    assert frac <= 0.01 and r_e2e < 2e-2, (r_e2e, frac)                  # final latents rel-L2 <= 2e-2
    p = np.array(panel)
    rec_small, edit_small = p[::4, 1024:1536:4], p[::4, 1536::4]
    d_rec = np.abs(rec_small.astype(np.int32) - g["recon_image_small"].astype(np.int32)).mean()
    d_edit = np.abs(edit_small.astype(np.int32) - g["edited_image_small"].astype(np.int32)).mean()
    ps_rec, ps_edit = psnr_u8(rec_small, g["recon_image_small"]), psnr_u8(edit_small, g["edited_image_small"])
    print("sd1 full width: inversion %.2e offsets %.2e recon %.2e edit %.2e (mask flips %.3f%%); panels mean|d| %.2f / %.2f, PSNR %.1f / %.1f dB"
          % (r_inv, r_nl, r_rec, r_edit, 100 * frac, d_rec, d_edit, ps_rec, ps_edit))
    assert d_rec <= 2.0 and d_edit <= 2.0, (d_rec, d_edit)               # decoded pixels mean |diff| <= 2/255
    assert ps_rec >= 35.0 and ps_edit >= 35.0, (ps_rec, ps_edit)         # PSNR(native, reference) >= 35 dB
    eng.close()


# ------------------------------------------------------------------------- (a2) full SD-1.x width AND the full 50 + 50-step schedule
def _subsampled_stage_errors(st, g, steps):
    xs = torch.stack([x.cpu() for x in st["x_stars"]])
    nl = torch.stack([x.cpu() for x in st["noise_loss_list"]])
    xi, ni = [int(i) for i in g["x_stars_index"]], [int(i) for i in g["noise_loss_index"]]
    gx, gn = torch.from_numpy(g["x_stars"]), torch.from_numpy(g["noise_loss"])
    inv = {k: rel(xs[k], gx[j]) for j, k in enumerate(xi)}
    # an offset is a small difference of two latents: its error is measured against the latent it corrects (x*_{t-1}), as in the other loops
    off = {k: ((nl[k] - gn[j]).norm() / xs[steps - k - 1].norm().clamp_min(1e-9) / 2 ** 0.5).item() for j, k in enumerate(ni)}
    return inv, off


@pytest.mark.parametrize("fixture", ["e2e_sd1_50.npz", "e2e_sd1_replace_50.npz"])
def test_sd1_full_width_full_schedule_against_reference_golden(fixture):
    """The benchmarked configuration at the benchmarked schedule: the reference's own P2PEditor("directinversion+p2p") at full SD-1.x
    width, 50 + 50 steps, Refine + Reweight + LocalBlend (tests/golden/e2e_sd1_50.npz: 63 min of the build container's CPU, every 10th
    inversion latent / offset kept) against the product's P2PEditor on the same image, prompts and weights.  Bars: SURVEY 8(d).
    e2e_sd1_replace_50.npz (round 6): the same with is_replace_controller=True -- BASELINE config 2's "P2P AttentionReplace" -- on the cake pair."""
    if not os.path.exists(os.path.join(GOLD, fixture)):
        pytest.skip("%s not generated (oracle/make_golden.py %s)" % (fixture, fixture[:-4]))
    g = np.load(os.path.join(GOLD, fixture))
    is_replace = bool(g["is_replace"])
    cfg, steps = SD1, int(g["steps"])
    assert steps == 50
    pipe = NativePipeline(cfg, max_unet_rows=12, max_vae_images=2, text_encoder=SyntheticTextEncoder(cfg.cross_dim, seed=7))
    seed = int(g["weight_seed"])
    pipe.load_state_dict(weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed))
    src, tgt = str(g["src"]), str(g["tgt"])
    w0, w1 = [str(x) for x in g["blend"]]
    ed = P2PEditor(["directinversion+p2p"], "cuda", num_ddim_steps=steps, pipeline=pipe)
    panel, st = ed.edit_image_directinversion(_cat_image(), src, tgt, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
                                              blend_word=((w0,), (w1,)), eq_params={"words": (w1,), "values": (2,)}, return_stages=True,
                                              is_replace_controller=is_replace)
    inv, off = _subsampled_stage_errors(st, g, steps)
    rec, out = st["reconstruct_latent"].cpu(), st["latents"].cpu()
    g_rec, g_out = torch.from_numpy(g["reconstruct_latent"]), torch.from_numpy(g["edited_latents"])
    r_src = rel(out[0], g_out[0])
    r_rec = rel(rec[1], g_rec[1])
    scale = max(1.0, float(g_out[1].pow(2).mean().sqrt()))
    r_out, frac = masked_rel(out[1], g_out[1], pix_tol=0.25 * scale)
    p = np.array(panel)
    rec_small, edit_small = p[::4, 1024:1536:4], p[::4, 1536::4]
    d = [float(np.abs(a.astype(np.int32) - b.astype(np.int32)).mean()) for a, b in ((rec_small, g["recon_image_small"]), (edit_small, g["edited_image_small"]))]
    ps = [float(psnr_u8(rec_small, g["recon_image_small"])), float(psnr_u8(edit_small, g["edited_image_small"]))]
    rep = {"inversion_rel_l2_at_step": inv, "offset_abs_err_over_latent_norm_at_step": off, "final_source_rel_l2": r_src,
           "final_reconstruction_target_rel_l2": r_rec, "final_edit_rel_l2": rel(out[1], g_out[1]), "final_edit_rel_l2_outside_localblend_flips": r_out,
           "localblend_mask_flip_fraction": frac, "edit_latent_rms": scale, "panel_mean_abs_diff": d, "panel_psnr_db": ps}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "drift_sd1_full_width_50%s.json" % ("_replace" if is_replace else "")), "w"), indent=1)
    print("sd1 full width 50+50 vs the reference:", json.dumps(rep))
    for k, r in inv.items():
        assert r < 4e-3 * max(k, 1) ** 0.5, (k, r)                       # DDIM latents after k steps: <= 4e-3 sqrt(k)
    assert max(off.values()) < 2e-2, off
    assert r_src < 2e-2, r_src                                           # direct inversion: the source branch returns to x*_0
    assert r_rec < 2e-2, r_rec
    assert frac <= 0.005 and r_out < 2e-2, (r_out, frac)                 # final latents after 50 + 50 steps
    assert max(d) <= 2.0 and min(ps) >= 35.0, (d, ps)                    # decoded pixels
    pipe.engine.close()


# ----------------------------------------------------------------------------------------------- (b) 50 + 50 steps vs the oracle
def oracle_tables(c, steps):
    """native controller (p2p/attention_control.py) -> the table dict of oracle.p2p_oracle.EditController"""
    inner = c.prev_controller if isinstance(c, ac.AttentionReweight) else c
    t = {"cross_alpha": c.cross_replace_alpha.reshape(steps + 1, 77), "self_range": c.num_self_replace,
         "equalizer": c.equalizer.reshape(-1) if isinstance(c, ac.AttentionReweight) else None, "lb": None}
    if isinstance(inner, ac.AttentionReplace):
        t.update(kind="replace", mapper=inner.mapper[0])
    elif isinstance(inner, ac.AttentionRefine):
        t.update(kind="refine", mapper=inner.mapper[0], alphas=inner.alphas.reshape(-1))
    else:
        t.update(kind="none")
    if c.local_blend is not None:
        t["lb"] = {"alpha_layers": c.local_blend.alpha_layers.reshape(2, 77), "start": c.local_blend.start_blend, "th": c.local_blend.th[0]}
    return t


@pytest.mark.parametrize("name", ["tiny16_replace_reweight", "small64lb_refine_reweight_localblend"])
def test_full_50_step_schedule_against_oracle(name):
    steps = 50
    if name.startswith("tiny16"):
        cfg, wseed, is_replace, blend = TINY16, 5, True, None
        prompts = ["a round cake with orange frosting on a wooden plate", "a square cake with orange frosting on a wooden plate"]
        eq = {"words": ("square",), "values": (2,)}
    else:
        # 64 x 64 latents (LocalBlend's 16 x 16 maps) with attention at the 32^2 / 16^2 / 8^2 levels only: the CPU oracle would spend
        # ten minutes materialising 4096 x 4096 self-attention tensors otherwise (the 64^2 sites are covered at 2-6 steps elsewhere)
        cfg, wseed, is_replace, blend = SMALL64_LB, 2, False, (("cat",), ("dog",))
        prompts = ["a cat sitting on a wooden chair", "a dog sitting on a wooden chair"]
        eq = {"words": ("dog",), "values": (2,)}
    usd, vsd = weights.unet_state_dict(cfg, wseed), weights.vae_state_dict(cfg, wseed)
    pipe = NativePipeline(cfg, max_unet_rows=12, max_vae_images=2, text_encoder=SyntheticTextEncoder(cfg.cross_dim, seed=3))
    pipe.load_state_dict(usd, vsd)
    eng = pipe.engine
    pipe.scheduler.set_timesteps(steps)
    ts = pipe.scheduler.timesteps.numpy()
    S = cfg.sample_size
    g = torch.Generator().manual_seed(31)
    z0 = torch.randn(1, 4, S, S, generator=g)
    ids = pipe.tokenizer([""] * 2 + prompts, padding="max_length", max_length=77, return_tensors="pt").input_ids
    ctx = pipe.text_encoder(ids)[0].float().cpu()                        # rows [unc_src, unc_tgt, cond_src, cond_tgt]
    ac_ = po.alphas_cumprod()
    torch.set_num_threads(min(32, os.cpu_count() or 1))

    def unet_fn(lat, t, c, hook):
        with torch.no_grad():
            return sd_oracle.unet_forward(usd, cfg, lat, t, c, hook)

    c = ac.make_controller(pipe, prompts, is_replace, {"default_": 0.4}, 0.6, blend, eq, num_ddim_steps=steps)
    tables = c.tables()
    assert tables.cross_alpha.shape == (51, 77) and tables.self_range == (0, 30)
    assert tables.cross_alpha[:20].min() == 1.0 and tables.cross_alpha[20:].max() == 0.0
    if blend is not None:
        assert tables.lb_start == 10

    # ---- oracle: the reference's phase order, fp32
    ref_lat = po.ddim_loop(unet_fn, z0, ctx[2:3], po.make_timesteps(steps), ac_, ac_[0])
    ref_nl = po.offset_calculate(unet_fn, ref_lat, ctx, po.make_timesteps(steps), ac_, ac_[0], 7.5)
    nsites = pipe.unet.num_att_layers
    ref_rec = po.guidance_forward(unet_fn, ref_lat[-1], ctx, ref_nl, po.StoreController(nsites), po.make_timesteps(steps), ac_, ac_[0], 7.5)
    ref_out = po.guidance_forward(unet_fn, ref_lat[-1], ctx, ref_nl, po.EditController(nsites, oracle_tables(c, steps)),
                                  po.make_timesteps(steps), ac_, ac_[0], 7.5)
    # ---- native: the product's schedule (B=1 inversion loop, then one 12-row lock-step launch per timestep), all on its OWN trajectory
    got_lat = eng.ddim_invert(z0, ctx[2:3], ts)
    nl, lats = eng.direct_edit(got_lat, ctx[None], [None, [tables]], ts, 7.5)
    ref_lat_t, ref_nl_t = torch.stack(ref_lat), torch.stack(ref_nl)
    drift = {"config": name, "steps": steps,
             "inversion_rel_l2_by_step": [rel(got_lat[i], ref_lat_t[i]) for i in range(1, steps + 1)],
             "offset_abs_err_over_latent_rms_by_step": [((nl[i, 0].cpu() - ref_nl_t[i]).norm() / ref_lat_t[steps - i - 1].norm().clamp_min(1e-9)).item()
                                                        for i in range(steps)]}
    r_inv = rel(got_lat, ref_lat_t)
    assert r_inv < 4e-3 * steps ** 0.5, r_inv                            # DDIM latents after k steps: <= 4e-3 sqrt(k)
    assert max(drift["inversion_rel_l2_by_step"]) < 4e-3 * steps ** 0.5
    # offsets are differences of two nearly equal latents: error relative to the latent scale (as in the 2-step tests)
    assert max(drift["offset_abs_err_over_latent_rms_by_step"]) < 2e-2, max(drift["offset_abs_err_over_latent_rms_by_step"])
    rec, out = lats[0, 0].cpu(), lats[1, 0].cpu()
    r_src = rel(out[0], z0[0])
    r_rec = rel(rec[1], ref_rec[1])
    scale = max(1.0, float(ref_out[1].pow(2).mean().sqrt()))            # LocalBlend flips are judged relative to the latent scale
    r_out, frac = masked_rel(out[1], ref_out[1], pix_tol=0.25 * scale)
    r_out_all = rel(out[1], ref_out[1])
    # decoded images: native VAE on native latents vs oracle VAE on oracle latents
    with torch.no_grad():
        ref_img = po.latent2image(lambda z: sd_oracle.vae_decode(vsd, cfg, z), torch.stack([ref_rec[0], ref_out[1]]))
    got_img = eng.latent2image(torch.stack([rec[0], out[1]])).cpu().numpy()
    d = [float(np.abs(got_img[i].astype(np.int32) - ref_img[i].astype(np.int32)).mean()) for i in range(2)]
    ps = [float(psnr_u8(got_img[i], ref_img[i])) for i in range(2)]
    # Conditioning of the edit branch itself, measured only when the stated bar is missed: the SAME fp32 oracle loop with each UNet
    # output perturbed by one fp16 rounding (relative 2^-11) -- far less than any fp16-storage pipeline incurs.  With random weights,
    # guidance 7.5 and the x2 equalizer the target branch is not a contraction, and the oracle itself can move by more than 2e-2.
    cond, ps_cond = None, None
    if not (r_out < 2e-2 and ps[1] >= 35.0):
        gen = torch.Generator().manual_seed(77)

        def unet_fn_rounded(lat, t, c_, hook):
            e = unet_fn(lat, t, c_, hook)
            return e * (1 + 2.0 ** -11 * torch.randn(e.shape, generator=gen))

        pert = po.guidance_forward(unet_fn_rounded, ref_lat[-1], ctx, ref_nl, po.EditController(nsites, oracle_tables(c, steps)),
                                   po.make_timesteps(steps), ac_, ac_[0], 7.5)
        cond = rel(pert[1], ref_out[1])
        with torch.no_grad():
            pert_img = po.latent2image(lambda z: sd_oracle.vae_decode(vsd, cfg, z), pert[1:2])
        ps_cond = float(psnr_u8(pert_img[0], ref_img[1]))
    drift.update(final_source_vs_z0=r_src, final_reconstruction_rel_l2=r_rec, final_edit_rel_l2=r_out_all,
                 final_edit_rel_l2_outside_localblend_flips=r_out, localblend_mask_flip_fraction=frac, edit_latent_rms=scale,
                 oracle_edit_rel_l2_under_one_fp16_rounding_per_step=cond)
    drift.update(image_mean_abs_diff=d, image_psnr_db=ps, oracle_edit_image_psnr_db_under_one_fp16_rounding_per_step=ps_cond)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(drift, open(os.path.join(ROOT, "gpurun_out", "drift_%s.json" % name), "w"), indent=1)
    print("50+50 steps %s: inversion %.2e (max step %.2e), offsets max %.2e, source %.2e, recon %.2e, edit %.2e (outside %.3f%% flips "
          "%.2e%s), images mean|d| %.2f / %.2f, PSNR %.1f / %.1f dB"
          % (name, r_inv, max(drift["inversion_rel_l2_by_step"]), max(drift["offset_abs_err_over_latent_rms_by_step"]), r_src, r_rec,
             r_out_all, 100 * frac, r_out, "" if cond is None else "; oracle under one fp16 rounding per step %.2e / %.1f dB" % (cond, ps_cond),
             d[0], d[1], ps[0], ps[1]))
    # SURVEY 8(d), 50 + 50 steps: final latents rel-L2 <= 2e-2, decoded images PSNR >= 35 dB / mean |diff| <= 2/255 -- asserted as
    # stated on the source branch (what direct inversion promises: x*_0 back) and the reconstruction pass's target row.
    assert r_src < 2e-2, r_src
    assert r_rec < 2e-2, r_rec
    assert torch.equal(rec[0], out[0])
    assert d[0] <= 2.0 and ps[0] >= 35.0, (d, ps)
    # The controller-edited target row: the same stated bars, asserted as stated.  (The conditioning figures above -- the fp32 oracle
    # re-run with one fp16 rounding per step -- are computed only when a bar is missed and are diagnostics in the drift file, not a bar.)
    assert frac <= 0.005, frac
    assert r_out < 2e-2, (r_out, cond)
    assert ps[1] >= 35.0 and d[1] <= 2.0, (ps, d, ps_cond)
    eng.close()


# ----------------------------------------------------------------------------------------------- (c) inserted target tokens
@pytest.mark.parametrize("name", ["insert2", "insert3"])
def test_refine_with_inserted_tokens_against_reference_golden(name):
    g = np.load(os.path.join(GOLD, "e2e_%s.npz" % name))
    cfg, steps = SMALL64, int(g["steps"])
    pipe = NativePipeline(cfg, max_unet_rows=12, max_vae_images=2, text_encoder=SyntheticTextEncoder(cfg.cross_dim, seed=7))
    pipe.load_state_dict(weights.unet_state_dict(cfg, 2), weights.vae_state_dict(cfg, 2))
    eng = pipe.engine
    pipe.scheduler.set_timesteps(steps)
    ts = pipe.scheduler.timesteps.numpy()
    ctx = torch.from_numpy(g["context"]).float()
    x_stars = torch.from_numpy(g["x_stars"])
    src, tgt = str(g["src"]), str(g["tgt"])
    w0, w1 = [str(x) for x in g["blend"]]
    use_blend = bool(g["use_blend"])
    ctrl = ac.make_controller(pipe, [src, tgt], False, {"default_": 0.4}, 0.6, ((w0,), (w1,)) if use_blend else None,
                              {"words": (w1,), "values": (2,)} if use_blend else None, num_ddim_steps=steps)
    inner = ctrl.prev_controller if use_blend else ctrl
    assert (inner.mapper[0] == -1).any() and (inner.alphas.reshape(-1) == 0).any()       # the case under test: inserted tokens
    nl, lats = eng.direct_edit(x_stars, ctx[None], [None, [ctrl.tables()]], ts, 7.5)
    assert rel(nl[:, 0], g["noise_loss"]) < 1.5e-2
    assert rel(lats[0, 0][1], g["reconstruct_latent"][1]) < 1.5e-2
    r, frac = masked_rel(lats[1, 0][1], torch.from_numpy(g["edited_latents"])[1])
    assert frac <= 0.005 and r < 1.5e-2, (r, frac)
    # the edit matters: the reconstruction pass's target row (no controller) is a different latent
    assert rel(lats[0, 0][1], g["edited_latents"][1]) > 3 * r
    ed = P2PEditor(["directinversion+p2p"], "cuda", num_ddim_steps=steps, pipeline=pipe)
    panel = ed("directinversion+p2p", _cat_image(), src, tgt, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
               blend_word=((w0,), (w1,)) if use_blend else None, eq_params={"words": (w1,), "values": (2,)} if use_blend else None)
    edit_small = np.array(panel)[::4, 1536::4]
    assert np.abs(edit_small.astype(np.int32) - g["edited_image_small"].astype(np.int32)).mean() < 4.0
    eng.close()

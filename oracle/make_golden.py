"""Generate tests/golden/* by running the REFERENCE's own code (imported from /root/reference through oracle/ref_shim.py) on
CPU with the seeded synthetic weights.  Run in the build container only:   python -m oracle.make_golden

Fixtures (the reference has no tests / golden vectors of its own for this path -- SURVEY.md 8c -- so these pin the oracle):
  host_tables.json     get_word_inds / refinement + replacement mappers / alpha schedule / equalizer / LocalBlend selectors
  unet_tiny.npz        my_diffusers UNet2DConditionModel + the reference's hooked attention, TINY16, B=4
  vae_tiny.npz         my_diffusers AutoencoderKL encode mean / decode, TINY16
  unet_sd1.npz         full-width SD-1.x UNet, B=1
  vae_sd1.npz          full-width SD-1.x VAE encode (256x256) / decode (16x16 latent)
  e2e_refine.npz       models/p2p_editor.py P2PEditor("directinversion+p2p") stage outputs, SMALL64, 2+2 steps,
                       AttentionRefine + AttentionReweight + LocalBlend (the PIE-Bench default controller)
  e2e_replace.npz      same with is_replace_controller=True (AttentionReplace), no blend / reweight
  e2e_insert2/3.npz    same as e2e_refine for prompt pairs whose target INSERTS tokens (refinement mapper -1 / alpha 0 entries)
  e2e_sd1.npz          same as e2e_refine at the FULL SD-1.x width (the benchmarked configuration), weight seed 0
  e2e_sd1_50.npz       the same at the benchmarked SCHEDULE too: full width, 50 + 50 steps (every 10th inversion latent / offset kept)
  e2e_variants.npz     the reference's P2PEditor run on six more method strings that share the loop (ddim+p2p,
                       negative-prompt-inversion+p2p, a vary-guidance, a not_full, a skip_step and the add-target ablation):
                       inversion latents / offsets where they differ from e2e_refine, reconstruction and edited latents
  e2e_masactrl.npz     run_editing_masactrl.py MasaCtrlEditor: directinversion+masactrl and ddim+masactrl stage outputs
  e2e_substruct.npz    AttentionRefine + LocalBlend(substruct_words=...) built from the reference's classes by hand, 4 + 4 steps
  e2e_proximal.npz     P2PEditor("negative-prompt-inversion+proximal-guidance") with the sweep script's arguments (l0) and l1
  e2e_proximal_recon.npz  the same method with use_reconstruction_guidance=True (masked pred-x0 pull + dilated edit mask), 4 steps
  e2e_null_text.npz    P2PEditor("null-text-inversion+p2p"): inversion latents, the optimised per-step unconditional embeddings, the loss
                       of every Adam iteration, reconstruction / edited latents (pins the oracle of the not-yet-built native path)
  null_text_family_tiny.npz  the edit passes of null-text-inversion+p2p / ..._single_branch+p2p / null-text-inversion+proximal-guidance on a
                       128 x 128 crop (TINY16 weights, AttentionRefine): per-step embeddings on every / the first unconditional row, l0 proximal step
  null_latent_tiny.npz DirectInversion.invert_null_latent (ablation_null-latent-inversion+p2p) on a 128 x 128 crop, TINY16 weights: inversion
                       latents, per-step latent offsets, every Adam iteration's loss
  e2e_null_text_sd1.npz / e2e_masactrl_sd1.npz / unet_ctxgrad_sd1.npz  BASELINE configs 4 and 5 at the FULL SD-1.x width (weight seed 0): the
                       reference's P2PEditor("null-text-inversion+p2p") with 2 steps x 10 Adam iterations, its MasaCtrlEditor with 4 steps (mutual
                       self-attention from step 1), and torch.autograd through its UNet for d eps / d context (one row)
  clip_tiny/sd1.npz    transformers CLIPTextModel last_hidden_state (the reference's model.text_encoder), seeded weights
  method_dispatch.json P2PEditor.__call__'s routing of its 39 method strings (handler + method-specific arguments)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from pnpinversion_amd import weights  # noqa: E402
from pnpinversion_amd.config import SD1, SMALL64, TINY16  # noqa: E402
from pnpinversion_amd.text import SyntheticTextEncoder, WordTokenizer  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

PROMPT_PAIRS = [
    ("a cat sitting on a wooden chair", "a dog sitting on a wooden chair", "cat", "dog"),
    ("a round cake with orange frosting on a wooden plate", "a square cake with orange frosting on a wooden plate", "cake", "cake"),
    ("a photograph of a mountain", "a watercolor photograph of a snowy mountain", "mountain", "mountain"),
    ("an elephant walking", "a strawberry elephant walking slowly", "elephant", "elephant"),
]


def host_tables():
    ref_shim.install()
    from models.p2p import seq_aligner
    from models.p2p.attention_control import LocalBlend, get_equalizer
    from utils.utils import get_time_words_attention_alpha, get_word_inds
    tok = WordTokenizer()
    out = []
    for (src, tgt, w0, w1) in PROMPT_PAIRS:
        e = {"src": src, "tgt": tgt, "blend": [w0, w1]}
        e["word_inds_src"] = get_word_inds(src, w0, tok).tolist()
        e["word_inds_tgt"] = get_word_inds(tgt, w1, tok).tolist()
        e["word_inds_int"] = get_word_inds(tgt, 1, tok).tolist()
        m, a = seq_aligner.get_refinement_mapper([src, tgt], tok)
        e["refine_mapper"] = m[0].tolist()
        e["refine_alphas"] = a[0].tolist()
        if len(src.split(" ")) == len(tgt.split(" ")):
            e["replace_mapper"] = seq_aligner.get_replacement_mapper([src, tgt], tok)[0].tolist()
        for steps in (50, 2):
            al = get_time_words_attention_alpha([src, tgt], steps, {"default_": 0.4}, tok)
            e["cross_alpha_%d" % steps] = al.reshape(steps + 1, 77).tolist()
        e["equalizer"] = get_equalizer(tgt, (w1,), (2,), tokenizer=tok)[0].tolist()
        with ref_shim.cuda_to_cpu():
            lb = LocalBlend([src, tgt], ((w0,), (w1,)), tokenizer=tok, device="cpu", num_ddim_steps=50)
        e["lb_alpha"] = lb.alpha_layers.reshape(2, 77).tolist()
        e["lb_start"] = lb.start_blend
        out.append(e)
    json.dump(out, open(os.path.join(OUT, "host_tables.json"), "w"))
    print("host_tables.json", len(out))


def model_goldens():
    ref_shim.install()
    from models.p2p.attention_control import register_attention_control
    for name, cfg, rows, t, seed in (("tiny", TINY16, 4, 500, 1), ("sd1", SD1, 1, 481, 0)):
        t0 = time.time()
        usd, vsd = weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed)
        g = torch.Generator().manual_seed(100 + seed)
        lat = torch.randn(rows, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
        ctx = weights.synth_context(cfg, rows, seed=200 + seed)
        unet = ref_shim.build_unet(cfg, usd)
        h = ref_shim._Holder()
        h.unet = unet
        register_attention_control(h, None)          # the reference's hooked attention forward with its DummyController
        with torch.no_grad():
            eps = unet(lat, torch.tensor(t), encoder_hidden_states=ctx)["sample"]
        np.savez_compressed(os.path.join(OUT, "unet_%s.npz" % name), latents=lat.numpy(), context=ctx.numpy().astype(np.float16),
                            t=np.int64(t), seed=np.int64(seed), eps=eps.numpy())
        del unet
        vae = ref_shim.build_vae(cfg, vsd)
        S = 64 if name == "tiny" else 256
        img = (torch.rand(1, 3, S, S, generator=g) * 2 - 1).half().float()   # fp16-representable (stored as fp16)
        zs = 8 if name == "tiny" else 16
        z = torch.randn(1, 4, zs, zs, generator=g)
        with torch.no_grad():
            mean = vae.encode(img)["latent_dist"].mean
            dec = vae.decode(z)["sample"]
        np.savez_compressed(os.path.join(OUT, "vae_%s.npz" % name), image=img.numpy().astype(np.float16), z=z.numpy(),
                            mean=mean.numpy(), dec=dec.numpy(), seed=np.int64(seed))
        del vae
        print("model goldens", name, "%.1fs" % (time.time() - t0))


def unet_ctxgrad_sd1():
    """d(eps . d_eps) / d(context) through the reference's own full-width UNet2DConditionModel with its hooked attention forward
    (torch.autograd, one row): what models/p2p/inversion.py:196-234 differentiates in every Adam iteration.  Pins pnpi_unet_context_grad
    (activation tape, dgrad through the forward GEMM family, flash attention backward) at the benchmarked width."""
    ref_shim.install()
    from models.p2p.attention_control import register_attention_control
    cfg, seed, t = SD1, 0, 481
    t0 = time.time()
    usd = weights.unet_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(300 + seed)
    lat = torch.randn(1, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = weights.synth_context(cfg, 1, seed=400 + seed).half().float()        # fp16-representable: the same values on both sides
    d_eps = torch.randn(1, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)
    unet = ref_shim.build_unet(cfg, usd)
    h = ref_shim._Holder()
    h.unet = unet
    register_attention_control(h, None)
    for q in unet.parameters():
        q.requires_grad_(False)
    cr = ctx.clone().requires_grad_(True)
    eps = unet(lat, torch.tensor(t), encoder_hidden_states=cr)["sample"]
    eps.backward(d_eps)
    np.savez_compressed(os.path.join(OUT, "unet_ctxgrad_sd1.npz"), latents=lat.numpy(), context=ctx.numpy().astype(np.float16), d_eps=d_eps.numpy(),
                        t=np.int64(t), seed=np.int64(seed), eps=eps.detach().numpy(), d_context=cr.grad.numpy())
    print("unet_ctxgrad_sd1 %.1fs |d_context| = %.3e" % (time.time() - t0, float(cr.grad.norm())))


def e2e(name, is_replace, blend, steps=2, cfg=SMALL64, seed=2, pair=None, keep_every=1):
    """keep_every > 1 (the 50 + 50-step full-width fixture): only every keep_every-th inversion latent / offset is stored
    (plus the last ones), so the committed file stays small.  pair: index into PROMPT_PAIRS (default: 1 for the Replace controller, which needs equal word counts, else 0).  Pairs 2 and
    3 insert tokens into the target prompt (seq_aligner.get_mapper's -1 / alpha 0 entries).  cfg=SD1 is the benchmarked width."""
    ref_shim.install()
    t0 = time.time()
    usd, vsd = weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed)
    tok = WordTokenizer()
    enc = SyntheticTextEncoder(cfg.cross_dim, seed=7)
    ed = ref_shim.build_editor(cfg, usd, vsd, tok, enc, steps)
    src, tgt, w0, w1 = PROMPT_PAIRS[pair if pair is not None else (1 if is_replace else 0)]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    stages = {}
    import models.p2p_editor as pe
    import models.p2p.inversion as inv
    orig_invert = inv.DirectInversion.invert

    def invert_spy(self, *a, **k):
        r = orig_invert(self, *a, **k)
        stages["x_stars"] = torch.stack([x.clone() for x in r[2]]).numpy()
        stages["noise_loss"] = torch.stack([x.clone() for x in r[3]]).numpy()
        stages["context"] = self.context.clone().numpy()
        return r

    orig_fwd = pe.direct_inversion_p2p_guidance_forward
    calls = []

    def fwd_spy(*a, **k):
        r = orig_fwd(*a, **k)
        calls.append(r[0].clone().numpy())
        return r

    inv.DirectInversion.invert = invert_spy
    pe.direct_inversion_p2p_guidance_forward = fwd_spy
    try:
        with ref_shim.cuda_to_cpu(), torch.no_grad():
            panel = ed("directinversion+p2p", image_path=img, prompt_src=src, prompt_tar=tgt, guidance_scale=7.5,
                       cross_replace_steps=0.4, self_replace_steps=0.6,
                       blend_word=((w0,), (w1,)) if blend else None,
                       eq_params={"words": (w1,), "values": (2,)} if blend else None,
                       is_replace_controller=is_replace)
    finally:
        inv.DirectInversion.invert = orig_invert
        pe.direct_inversion_p2p_guidance_forward = orig_fwd
    panel = np.array(panel)
    S = 512
    kept = {}
    if keep_every > 1:
        # x_stars[j] for j in xs_idx (always the end points); noise_loss[i] for i in nl_idx
        xs_idx = sorted(set(list(range(0, steps + 1, keep_every)) + [steps]))
        nl_idx = sorted(set(list(range(0, steps, keep_every)) + [steps - 1]))
        stages["x_stars"] = stages["x_stars"][xs_idx]
        stages["noise_loss"] = stages["noise_loss"][nl_idx]
        kept = dict(x_stars_index=np.array(xs_idx, np.int64), noise_loss_index=np.array(nl_idx, np.int64))
    np.savez_compressed(os.path.join(OUT, "e2e_%s.npz" % name), x_stars=stages["x_stars"], noise_loss=stages["noise_loss"], **kept,
                        context=stages["context"].astype(np.float16), reconstruct_latent=calls[0], edited_latents=calls[1],
                        recon_image_small=panel[::4, 2 * S:3 * S:4], edited_image_small=panel[::4, 3 * S::4],
                        src=src, tgt=tgt, blend=np.array([w0, w1]), steps=np.int64(steps), is_replace=np.bool_(is_replace),
                        use_blend=np.bool_(blend), weight_seed=np.int64(seed))
    print("e2e", name, "%.1fs" % (time.time() - t0))


def local_blend_substruct(steps=4, th=(0.3, 0.6)):
    """LocalBlend(substruct_words=...) (attention_control.py:97-118,134-143): no shipped script passes substruct_words (make_controller has
    no argument for them), so the controller is built by hand from the reference's classes -- AttentionRefine with
    LocalBlend([src, tgt], (("cat",), ("dog",)), substruct_words=(("chair",), ("chair",)), th=th) -- and run through the reference's
    DirectInversion.invert + direct_inversion_p2p_guidance_forward (models/p2p_editor.py:430-472), SMALL64, 4 + 4 steps.  th[1] is chosen so
    that the substruct mask is neither empty nor full on the seeded weights (the fractions of every get_mask call are stored)."""
    ref_shim.install()
    t0 = time.time()
    cfg, seed = SMALL64, 2
    usd, vsd = weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    model = ed.ldm_stable
    src, tgt, w0, w1 = PROMPT_PAIRS[0]
    sub = "chair"
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    from models.p2p.inversion import DirectInversion
    from models.p2p.p2p_guidance_forward import direct_inversion_p2p_guidance_forward
    from models.p2p.attention_control import AttentionRefine, LocalBlend
    from utils.utils import load_512
    fracs = []
    orig_get_mask = LocalBlend.get_mask

    def get_mask_spy(self, maps, alpha, use_pool):
        m = orig_get_mask(self, maps, alpha, use_pool)
        fracs.append([float(use_pool), float(m[1].float().mean())])
        return m

    LocalBlend.get_mask = get_mask_spy
    try:
        with ref_shim.cuda_to_cpu(), torch.no_grad():
            inv = DirectInversion(model=model, num_ddim_steps=steps)
            _, _, x_stars, noise_loss = inv.invert(image_gt=load_512(img), prompt=[src, tgt], guidance_scale=7.5)
            lb = LocalBlend([src, tgt], ((w0,), (w1,)), substruct_words=((sub,), (sub,)), th=th, tokenizer=model.tokenizer, device="cpu",
                            num_ddim_steps=steps)
            ctrl = AttentionRefine([src, tgt], steps, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, local_blend=lb,
                                   tokenizer=model.tokenizer, device="cpu")
            latents, _ = direct_inversion_p2p_guidance_forward(model=model, prompt=[src, tgt], controller=ctrl, noise_loss_list=noise_loss,
                                                               latent=x_stars[-1], num_inference_steps=steps, guidance_scale=7.5,
                                                               generator=None)
            # the same edit without the substruct words: what the substruct mask changes
            lb0 = LocalBlend([src, tgt], ((w0,), (w1,)), th=th, tokenizer=model.tokenizer, device="cpu", num_ddim_steps=steps)
            ctrl0 = AttentionRefine([src, tgt], steps, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, local_blend=lb0,
                                    tokenizer=model.tokenizer, device="cpu")
            nfr = len(fracs)
            latents0, _ = direct_inversion_p2p_guidance_forward(model=model, prompt=[src, tgt], controller=ctrl0, noise_loss_list=noise_loss,
                                                                latent=x_stars[-1], num_inference_steps=steps, guidance_scale=7.5,
                                                                generator=None)
    finally:
        LocalBlend.get_mask = orig_get_mask
    np.savez_compressed(os.path.join(OUT, "e2e_substruct.npz"), x_stars=torch.stack(list(x_stars)).numpy(),
                        noise_loss=torch.stack(list(noise_loss)).numpy(), context=inv.context.numpy().astype(np.float16),
                        edited_latents=latents.numpy(), edited_latents_no_substruct=latents0.numpy(),
                        mask_fractions=np.array(fracs[:nfr], np.float64), src=src, tgt=tgt, blend=np.array([w0, w1]), substruct=np.array([sub, sub]),
                        th=np.array(th, np.float64), steps=np.int64(steps), weight_seed=np.int64(seed))
    print("local_blend_substruct %.1fs; get_mask (use_pool, fraction of the target mask set):" % (time.time() - t0), fracs[:nfr],
          "| rel diff to the edit without substruct words %.3e" % float((latents[1] - latents0[1]).norm() / latents0[1].norm()))


VARIANT_METHODS = ["ddim+p2p", "negative-prompt-inversion+p2p", "directinversion+p2p_guidance_25_5",
                   "ablation_directinversion_04+p2p", "ablation_directinversion_interval_2+p2p",
                   "ablation_directinversion_add-target+p2p"]


def variants(steps=2):
    """Same image / prompts / weights / controller settings as e2e_refine, other method strings of P2PEditor.__call__."""
    ref_shim.install()
    cfg = SMALL64
    usd, vsd = weights.unet_state_dict(cfg, 2), weights.vae_state_dict(cfg, 2)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    src, tgt, w0, w1 = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    import models.p2p_editor as pe
    import models.p2p.inversion as inv
    out = {"methods": np.array(VARIANT_METHODS), "steps": np.int64(steps), "src": src, "tgt": tgt, "blend": np.array([w0, w1])}
    fwd_names = ["direct_inversion_p2p_guidance_forward", "direct_inversion_p2p_guidance_forward_add_target", "p2p_guidance_forward",
                 "proximal_guidance_forward"]
    inv_names = ["invert", "invert_with_guidance_scale_vary_guidance", "invert_not_full", "invert_skip_step"]
    for m in VARIANT_METHODS:
        t0 = time.time()
        calls, stages = [], {}
        saved = {n: getattr(pe, n) for n in fwd_names}
        saved_inv = {n: getattr(inv.DirectInversion, n) for n in inv_names}

        def spy_fwd(f):
            def g(*a, **k):
                r = f(*a, **k)
                calls.append(r[0].clone().numpy())
                return r
            return g

        def spy_inv(f):
            def g(self, *a, **k):
                r = f(self, *a, **k)
                stages["x_stars"] = torch.stack([x.clone() for x in r[2]]).numpy()
                stages["noise_loss"] = torch.stack([x.clone() for x in r[3]]).numpy()
                return r
            return g

        for n in fwd_names:
            setattr(pe, n, spy_fwd(saved[n]))
        for n in inv_names:
            setattr(inv.DirectInversion, n, spy_inv(saved_inv[n]))
        try:
            with ref_shim.cuda_to_cpu(), torch.no_grad():
                panel = ed(m, image_path=img, prompt_src=src, prompt_tar=tgt, guidance_scale=7.5, cross_replace_steps=0.4,
                           self_replace_steps=0.6, blend_word=((w0,), (w1,)), eq_params={"words": (w1,), "values": (2,)},
                           is_replace_controller=False)
        finally:
            for n in fwd_names:
                setattr(pe, n, saved[n])
            for n in inv_names:
                setattr(inv.DirectInversion, n, saved_inv[n])
        assert len(calls) == 2, (m, len(calls))
        out[m + "/reconstruct_latent"] = calls[0]
        out[m + "/edited_latents"] = calls[1]
        if m.startswith("directinversion+p2p_guidance"):
            out[m + "/x_stars"] = stages["x_stars"]
        if "noise_loss" in stages and m != "ablation_directinversion_add-target+p2p":
            out[m + "/noise_loss"] = stages["noise_loss"]
        out[m + "/edited_image_small"] = np.array(panel)[::4, 3 * 512::4]
        print("variant", m, "%.1fs" % (time.time() - t0), {k: v.shape for k, v in out.items() if k.startswith(m + "/")})
    np.savez_compressed(os.path.join(OUT, "e2e_variants.npz"), **out)


ALL_METHODS = (["ddim+p2p", "null-text-inversion+p2p", "null-text-inversion+p2p_a800", "null-text-inversion+p2p_3090",
                "ablation_null-text-inversion_single_branch+p2p", "negative-prompt-inversion+p2p", "directinversion+p2p"] +
               ["directinversion+p2p_guidance_%s_%s" % (a, b) for a in ("0", "1", "25", "5", "75") for b in ("1", "5", "25", "75")] +
               ["null-text-inversion+proximal-guidance", "negative-prompt-inversion+proximal-guidance",
                "ablation_null-latent-inversion+p2p", "ablation_directinversion_08+p2p", "ablation_directinversion_04+p2p"] +
               ["ablation_directinversion_interval_%d+p2p" % k for k in (2, 5, 10, 24, 49)] +
               ["ablation_directinversion_add-target+p2p", "ablation_directinversion_add-source+p2p"])


def dispatch():
    """Which edit_image_* method the reference's P2PEditor.__call__ (models/p2p_editor.py:28-135) routes each of its 39 method
    strings to, and with which method-specific arguments (run_editing_p2p.py's call arguments)."""
    ref_shim.install()
    import models.p2p_editor as pe
    ed = pe.P2PEditor.__new__(pe.P2PEditor)
    rec = {}
    names = [n for n in dir(pe.P2PEditor) if n.startswith("edit_image")]
    for n in names:
        def mk(n):
            def f(*a, **k):
                keep = {kk: (vv if not isinstance(vv, float) else float(vv)) for kk, vv in k.items()
                        if kk in ("guidance_scale", "inverse_guidance_scale", "forward_guidance_scale", "scale", "skip_step", "proximal",
                                  "quantile", "recon_lr", "recon_t", "use_inversion_guidance", "use_reconstruction_guidance", "dilate_mask")}
                return (n, keep)
            return f
        setattr(ed, n, mk(n))
    for m in ALL_METHODS:
        r = ed(m, image_path="x", prompt_src="a", prompt_tar="b", guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
               blend_word=None, eq_params=None, proximal="l0", quantile=0.75, use_inversion_guidance=True, recon_lr=1, recon_t=400)
        rec[m] = {"handler": r[0], "kwargs": r[1]}
    try:
        ed("no-such-method", image_path="x", prompt_src="a", prompt_tar="b")
    except NotImplementedError as e:
        rec["__unknown__"] = str(e)
    json.dump(rec, open(os.path.join(OUT, "method_dispatch.json"), "w"), indent=1, sort_keys=True)
    print("method_dispatch.json", len(rec))


def clip_text():
    """transformers CLIPTextModel (the class the reference's pipeline instantiates as model.text_encoder) with the seeded weights
    of weights.clip_state_dict: last_hidden_state for two prompts, reduced (TINY16's) and SD-1.x (ViT-L/14 text tower) sizes."""
    from transformers import CLIPTextConfig, CLIPTextModel
    tok = WordTokenizer()
    ids = tok([PROMPT_PAIRS[0][0], PROMPT_PAIRS[2][1]], padding="max_length", max_length=77, return_tensors="pt").input_ids
    for name, cfg, seed in (("tiny", TINY16, 4), ("sd1", SD1, 0)):
        hc = CLIPTextConfig(vocab_size=cfg.clip_vocab, hidden_size=cfg.cross_dim, intermediate_size=cfg.clip_intermediate,
                            num_hidden_layers=cfg.clip_layers, num_attention_heads=cfg.clip_heads,
                            max_position_embeddings=cfg.ctx_len, hidden_act="quick_gelu")
        m = CLIPTextModel(hc).eval()
        sd = weights.clip_state_dict(cfg, seed)
        own = m.state_dict()
        pre = "text_model." if any(k.startswith("text_model.") for k in own) else ""
        missing, unexpected = m.load_state_dict({pre + k: v for k, v in sd.items()}, strict=False)
        assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
        with torch.no_grad():
            out = m(ids)[0]
        np.savez_compressed(os.path.join(OUT, "clip_%s.npz" % name), input_ids=ids.numpy().astype(np.int32), hidden=out.numpy(),
                            seed=np.int64(seed))
        print("clip", name, tuple(out.shape), float(out.std()))


def proximal(steps=2):
    """P2PEditor("negative-prompt-inversion+proximal-guidance") with the arguments run_editing_p2p.py:286-300 passes
    (proximal="l0", quantile=0.75, use_inversion_guidance=True, recon_lr=1, recon_t=400), and the 'l1' variant."""
    ref_shim.install()
    cfg = SMALL64
    usd, vsd = weights.unet_state_dict(cfg, 2), weights.vae_state_dict(cfg, 2)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    src, tgt, w0, w1 = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    import models.p2p_editor as pe
    out = {"steps": np.int64(steps), "src": src, "tgt": tgt, "blend": np.array([w0, w1])}
    for prox in ("l0", "l1"):
        calls = []
        saved = pe.proximal_guidance_forward

        def spy(*a, **k):
            r = saved(*a, **k)
            calls.append(r[0].clone().numpy())
            return r

        pe.proximal_guidance_forward = spy
        try:
            with ref_shim.cuda_to_cpu(), torch.no_grad():
                panel = ed("negative-prompt-inversion+proximal-guidance", image_path=img, prompt_src=src, prompt_tar=tgt,
                           guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6, blend_word=((w0,), (w1,)),
                           eq_params={"words": (w1,), "values": (2,)}, proximal=prox, quantile=0.75, use_inversion_guidance=True,
                           recon_lr=1, recon_t=400)
        finally:
            pe.proximal_guidance_forward = saved
        assert len(calls) == 2
        out[prox + "/reconstruct_latent"], out[prox + "/edited_latents"] = calls
        out[prox + "/edited_image_small"] = np.array(panel)[::4, 3 * 512::4]
        print("proximal", prox, {k: v.shape for k, v in out.items() if k.startswith(prox + "/")})
    np.savez_compressed(os.path.join(OUT, "e2e_proximal.npz"), **out)


def proximal_recon(steps=4):
    """P2PEditor("negative-prompt-inversion+proximal-guidance") with use_reconstruction_guidance=True (models/p2p_editor.py:324-413 ->
    proximal_guidance_forward.py:48-51,60-72 -> DDIMSchedulerDev.step's ref_image branch, scheduler_dev.py:68-76): 4 steps
    (t = 750, 500, 250, 0: the pull towards the encoded source image is active at t < recon_t = 400), dilate_mask = 1, l0 and l1."""
    ref_shim.install()
    cfg = SMALL64
    usd, vsd = weights.unet_state_dict(cfg, 2), weights.vae_state_dict(cfg, 2)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    src, tgt, w0, w1 = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    import models.p2p_editor as pe
    out = {"steps": np.int64(steps), "src": src, "tgt": tgt, "blend": np.array([w0, w1]), "recon_lr": np.float32(0.1), "recon_t": np.int64(400),
           "dilate_mask": np.int64(1)}
    import models.p2p.inversion as inv
    for prox in ("l0", "l1"):
        calls = []
        saved = pe.proximal_guidance_forward
        saved_inv = inv.NegativePromptInversion.invert

        def spy(*a, **k):
            r = saved(*a, **k)
            calls.append(r[0].clone().numpy())
            return r

        def spy_inv(self, *a, **k):
            r = saved_inv(self, *a, **k)
            out["image_enc_latent"] = r[1].clone().numpy()
            out["x_stars"] = torch.stack([x.clone() for x in r[2]]).numpy()
            out["context"] = self.context.clone().numpy().astype(np.float16)
            return r

        pe.proximal_guidance_forward = spy
        inv.NegativePromptInversion.invert = spy_inv
        try:
            with ref_shim.cuda_to_cpu(), torch.no_grad():
                panel = ed("negative-prompt-inversion+proximal-guidance", image_path=img, prompt_src=src, prompt_tar=tgt,
                           guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6, blend_word=((w0,), (w1,)),
                           eq_params={"words": (w1,), "values": (2,)}, proximal=prox, quantile=0.75, use_reconstruction_guidance=True,
                           recon_lr=0.1, recon_t=400, dilate_mask=1)
        finally:
            pe.proximal_guidance_forward = saved
            inv.NegativePromptInversion.invert = saved_inv
        assert len(calls) == 2
        out[prox + "/reconstruct_latent"], out[prox + "/edited_latents"] = calls
        out[prox + "/edited_image_small"] = np.array(panel)[::4, 3 * 512::4]
        print("proximal_recon", prox, {k: v.shape for k, v in out.items() if k.startswith(prox + "/")})
    np.savez_compressed(os.path.join(OUT, "e2e_proximal_recon.npz"), **out)


def proximal_inv_guidance(steps=4):
    """Inversion guidance of proximal_guidance_forward (models/p2p/proximal_guidance_forward.py:73-75), which no reference editor switches on
    (p2p_editor.py:368,593 pass inversion_guidance=False): the reference's own function, reached through its own editor
    ("negative-prompt-inversion+proximal-guidance", use_inversion_guidance=True -> recon_lr / recon_t / x_stars are passed on) with
      pos: inversion_guidance=True injected into the edit-stage call, recon_t = 400 (pull active at t = 250, 0 of the 4 steps), recon_lr 0.5
      neg: NO injection, recon_t = -600 in the edit-stage call: by the operator precedence of :73 the pull runs at t > 600 (t = 750) whatever
           the flag says (the editor itself cannot pass a negative recon_t: its reconstruction pass then dies on `1 - None`, :78)
      off: the same call without either (the effect under test must be resolved against it)."""
    ref_shim.install()
    cfg = SMALL64
    usd, vsd = weights.unet_state_dict(cfg, 2), weights.vae_state_dict(cfg, 2)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    src, tgt, w0, w1 = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    import models.p2p_editor as pe
    out = {"steps": np.int64(steps), "src": src, "tgt": tgt, "blend": np.array([w0, w1]), "recon_lr": np.float32(0.5), "dilate_mask": np.int64(1)}
    for name, recon_t, inject in (("pos", 400, True), ("neg", -600, False), ("off", 400, False)):
        calls = []
        saved = pe.proximal_guidance_forward

        def spy(*a, **k):
            if k.get("edit_stage") and k.get("prox") is not None:
                if inject:
                    k["inversion_guidance"] = True
                k["recon_t"] = recon_t          # (a negative recon_t crashes the editor's reconstruction pass -- `1 - None`, :78: edit stage only)
            r = saved(*a, **k)
            calls.append(r[0].clone().numpy())
            return r

        pe.proximal_guidance_forward = spy
        try:
            with ref_shim.cuda_to_cpu(), torch.no_grad():
                panel = ed("negative-prompt-inversion+proximal-guidance", image_path=img, prompt_src=src, prompt_tar=tgt,
                           guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6, blend_word=((w0,), (w1,)),
                           eq_params={"words": (w1,), "values": (2,)}, proximal="l0", quantile=0.75, use_inversion_guidance=True,
                           recon_lr=0.5, recon_t=400, dilate_mask=1)
        finally:
            pe.proximal_guidance_forward = saved
        assert len(calls) == 2
        out[name + "/recon_t"] = np.int64(recon_t)
        out[name + "/reconstruct_latent"], out[name + "/edited_latents"] = calls
        out[name + "/edited_image_small"] = np.array(panel)[::4, 3 * 512::4]
        print("proximal_inv_guidance", name, float(np.abs(calls[1]).mean()))
    d = lambda a, b: float(np.linalg.norm(out[a + "/edited_latents"] - out[b + "/edited_latents"]) / np.linalg.norm(out[b + "/edited_latents"]))
    print("pull on vs off: pos %.3e, neg %.3e" % (d("pos", "off"), d("neg", "off")))
    np.savez_compressed(os.path.join(OUT, "proximal_inv_guidance.npz"), **out)


def null_text(steps=3, cfg=SMALL64, seed=2, name="e2e_null_text", keep_every=1, perturb=None):
    """keep_every > 1 (round 6, e2e_null_text_sd1_20: 20 steps x 10 Adam iterations at full width): only every keep_every-th inversion
    latent / optimised embedding (plus the end points) is stored.  perturb = (relative sigma, seed): the SAME reference run with every UNet
    output multiplied by 1 + sigma * N(0, 1) (sigma = 2^-11: one fp16 rounding per UNet call, far less than any fp16-storage pipeline
    incurs) -- the reference's own sensitivity at this schedule, stored beside the golden as the yardstick of the latent bar
    (tests/test_gpu_sd1_configs.py), as tests/test_gpu_headline_parity.py does for the headline against the oracle.
    cfg=SD1 (name e2e_null_text_sd1, 2 steps x 10 Adam iterations, weight seed 0): BASELINE config 4 at the benchmarked width.
    P2PEditor("null-text-inversion+p2p") of the reference (models/p2p_editor.py:199-259): NullInversion.invert = ddim_inversion +
    null_optimization (10 Adam iterations per step through the UNet w.r.t. the 77 x D unconditional embedding, inversion.py:196-234),
    then p2p_guidance_forward twice with the per-step embeddings.  Same image / prompts / weights / controller as e2e_refine.
    The native path does not build this method yet (it needs the UNet backward pass); the fixture pins the ORACLE's restatement
    (oracle/p2p_oracle.py: null_optimization) so that the device implementation has a checker waiting."""
    ref_shim.install()
    usd, vsd = weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    src, tgt, w0, w1 = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")))[:, :, :3]
    import models.p2p_editor as pe
    import models.p2p.inversion as inv
    t0 = time.time()
    calls, stages, losses = [], {}, []
    orig_fwd, orig_inv = pe.p2p_guidance_forward, inv.NullInversion.invert
    orig_mse = inv.nnf.mse_loss

    def spy_fwd(*a, **k):
        r = orig_fwd(*a, **k)
        calls.append(r[0].clone().numpy())
        return r

    def spy_inv(self, *a, **k):
        r = orig_inv(self, *a, **k)
        stages["x_stars"] = torch.stack([x.clone() for x in r[2]]).numpy()
        stages["uncond"] = torch.stack([x.clone() for x in r[3]]).numpy()
        stages["context"] = self.context.clone().numpy()
        return r

    def spy_mse(*a, **k):
        r = orig_mse(*a, **k)
        losses.append(float(r.detach()))
        return r

    pe.p2p_guidance_forward, inv.NullInversion.invert, inv.nnf.mse_loss = spy_fwd, spy_inv, spy_mse
    hook = None
    if perturb is not None:
        sigma, pseed = perturb
        gen = torch.Generator().manual_seed(pseed)

        def noisy(_m, _a, out):
            e = out["sample"]
            out["sample"] = e * (1 + sigma * torch.randn(e.shape, generator=gen).to(e.dtype))
            return out

        hook = ed.ldm_stable.unet.register_forward_hook(noisy)
    try:
        with ref_shim.cuda_to_cpu():           # NOT under no_grad: null_optimization differentiates through the UNet
            panel = ed("null-text-inversion+p2p", image_path=img, prompt_src=src, prompt_tar=tgt, guidance_scale=7.5, cross_replace_steps=0.4,
                       self_replace_steps=0.6, blend_word=((w0,), (w1,)), eq_params={"words": (w1,), "values": (2,)},
                       is_replace_controller=False)
    finally:
        pe.p2p_guidance_forward, inv.NullInversion.invert, inv.nnf.mse_loss = orig_fwd, orig_inv, orig_mse
        if hook is not None:
            hook.remove()
    assert len(calls) == 2
    kept = {}
    if keep_every > 1:
        xs_idx = sorted(set(list(range(0, steps + 1, keep_every)) + [steps]))
        un_idx = sorted(set(list(range(0, steps, keep_every)) + [steps - 1]))
        stages["x_stars"], stages["uncond"] = stages["x_stars"][xs_idx], stages["uncond"][un_idx]
        kept = dict(x_stars_index=np.array(xs_idx, np.int64), uncond_index=np.array(un_idx, np.int64))
    if perturb is not None:
        kept.update(perturb_sigma=np.float64(perturb[0]), perturb_seed=np.int64(perturb[1]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x_stars=stages["x_stars"], uncond_embeddings=stages["uncond"], **kept,
                        context=stages["context"], losses=np.array(losses, dtype=np.float64), reconstruct_latent=calls[0],
                        edited_latents=calls[1], edited_image_small=np.array(panel)[::4, 3 * 512::4], steps=np.int64(steps), src=src, tgt=tgt,
                        blend=np.array([w0, w1]), weight_seed=np.int64(seed))
    print("null_text %.1fs, %d inner iterations, loss %.3e -> %.3e" % (time.time() - t0, len(losses), losses[0], losses[-1]))


def null_latent(steps=3):
    """DirectInversion.invert_null_latent (inversion.py:418-470; the inversion of "ablation_null-latent-inversion+p2p") on a 128 x 128
    crop (16 x 16 latents: the B = 2 optimisation through 4096-token attention maps would take minutes per step on CPU; the guidance
    passes that consume the offsets are direct_inversion_p2p_guidance_forward, pinned by e2e_refine).  Pins p2p_oracle.null_latent_calculate."""
    ref_shim.install()
    cfg = TINY16
    usd, vsd = weights.unet_state_dict(cfg, 1), weights.vae_state_dict(cfg, 1)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    src, tgt, _, _ = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")).convert("RGB").resize((128, 128)))
    import models.p2p.inversion as inv
    losses = []
    orig_mse = inv.nnf.mse_loss

    def spy_mse(*a, **k):
        r = orig_mse(*a, **k)
        losses.append(float(r.detach()))
        return r

    inv.nnf.mse_loss = spy_mse
    t0 = time.time()
    try:
        with ref_shim.cuda_to_cpu():
            di = inv.DirectInversion(model=ed.ldm_stable, num_ddim_steps=steps)
            _, _, x_stars, nl = di.invert_null_latent(image_gt=img, prompt=[src, tgt], guidance_scale=7.5)
    finally:
        inv.nnf.mse_loss = orig_mse
    np.savez_compressed(os.path.join(OUT, "null_latent_tiny.npz"), x_stars=torch.stack([x.detach() for x in x_stars]).numpy(),
                        noise_loss=torch.stack([x.detach() for x in nl]).numpy(), context=di.context.detach().numpy(),
                        losses=np.array(losses, dtype=np.float64), steps=np.int64(steps), src=src, tgt=tgt)
    print("null_latent %.1fs, %d inner iterations, |offset| %.3e" % (time.time() - t0, len(losses), float(torch.stack(nl).abs().mean())))


def null_text_family(steps=3):
    """The other consumers of null-text embeddings, on a 128 x 128 crop with TINY16 weights (16 x 16 latents; AttentionRefine without
    LocalBlend, whose 16 x 16 attention maps do not exist at this size; the loops' hard-coded 512 in the init_latent call is overridden by the
    latent's own size): NullInversion.invert, then the edit pass of
      "null-text-inversion+p2p"                          p2p_guidance_forward               (the embedding on every unconditional row)
      "ablation_null-text-inversion_single_branch+p2p"   p2p_guidance_forward_single_branch (on the first row only)
      "null-text-inversion+proximal-guidance"            proximal_guidance_forward with the sweep script's arguments (l0, quantile 0.75)."""
    ref_shim.install()
    cfg = TINY16
    usd, vsd = weights.unet_state_dict(cfg, 1), weights.vae_state_dict(cfg, 1)
    ed = ref_shim.build_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    model = ed.ldm_stable
    src, tgt, _, _ = PROMPT_PAIRS[0]
    from PIL import Image
    img = np.array(Image.open(os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")).convert("RGB").resize((128, 128)))
    from models.p2p.inversion import NullInversion
    from models.p2p.p2p_guidance_forward import p2p_guidance_forward, p2p_guidance_forward_single_branch
    from models.p2p.proximal_guidance_forward import proximal_guidance_forward
    from models.p2p.attention_control import make_controller
    import models.p2p.p2p_guidance_forward as gf
    import models.p2p.proximal_guidance_forward as pf
    orig_init = gf.init_latent

    def init_latent_any_size(latent, model_, height, width, generator, batch_size):
        # the guidance loops hard-code height = width = 512 for utils.init_latent's expand(); everything else in them is size-agnostic
        return orig_init(latent, model_, 8 * latent.shape[-2], 8 * latent.shape[-1], generator, batch_size)

    gf.init_latent = pf.init_latent = init_latent_any_size
    t0 = time.time()
    out = {"steps": np.int64(steps), "src": src, "tgt": tgt}
    with ref_shim.cuda_to_cpu():
        ni = NullInversion(model=model, num_ddim_steps=steps)
        _, _, x_stars, unc = ni.invert(image_gt=img, prompt=src, guidance_scale=7.5)
        out["x_stars"] = torch.stack([x.detach() for x in x_stars]).numpy()
        out["uncond_embeddings"] = torch.stack([u.detach() for u in unc]).numpy()
        out["context"] = ni.context.detach().numpy()

        def ctrl():
            return make_controller(pipeline=model, prompts=[src, tgt], is_replace_controller=False, cross_replace_steps={"default_": 0.4},
                                   self_replace_steps=0.6, blend_words=None, equilizer_params=None, num_ddim_steps=steps, device="cpu")
        x_t = x_stars[-1]
        out["p2p/edited_latents"] = p2p_guidance_forward(model=model, prompt=[src, tgt], controller=ctrl(), latent=x_t, num_inference_steps=steps,
                                                         guidance_scale=7.5, generator=None, uncond_embeddings=unc)[0].numpy()
        out["single_branch/edited_latents"] = p2p_guidance_forward_single_branch(
            model=model, prompt=[src, tgt], controller=ctrl(), latent=x_t, num_inference_steps=steps, guidance_scale=7.5, generator=None,
            uncond_embeddings=unc)[0].numpy()
        out["proximal/edited_latents"] = proximal_guidance_forward(
            model=model, prompt=[src, tgt], controller=ctrl(), latent=x_t, guidance_scale=7.5, generator=None, uncond_embeddings=unc,
            edit_stage=True, prox="l0", quantile=0.75, image_enc=None, recon_lr=1, recon_t=400, x_stars=x_stars, dilate_mask=1)[0].numpy()
    gf.init_latent = pf.init_latent = orig_init
    np.savez_compressed(os.path.join(OUT, "null_text_family_tiny.npz"), **out)
    print("null_text_family %.1fs" % (time.time() - t0), {k: getattr(v, "shape", v) for k, v in out.items()})


def masactrl(steps=6, start_step=2, start_layer=10, cfg=SMALL64, seed=2, name="e2e_masactrl", keep_every=1, perturb=None,
             methods=("directinversion+masactrl", "ddim+masactrl")):
    """cfg=SD1 (name e2e_masactrl_sd1, 4 steps, mutual self-attention from step 1, weight seed 0): BASELINE config 5 at the benchmarked width.
    run_editing_masactrl.py MasaCtrlEditor("directinversion+masactrl" / "ddim+masactrl"), SMALL64, 6 steps, mutual
    self-attention from step 2 in transformer blocks 10..15.  The pipeline's __call__ defaults to 50 sampling steps and the
    editor relies on that default for its second call (run_editing_masactrl.py:118-121); the default is set to `steps` here."""
    import inspect
    usd, vsd = weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed)
    ed = ref_shim.build_masactrl_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    from models.masactrl.diffuser_utils import MasaCtrlPipeline
    import models.p2p.inversion as inv
    params = list(inspect.signature(MasaCtrlPipeline.__call__).parameters.values())
    names = [p.name for p in params if p.default is not inspect.Parameter.empty]
    d = list(MasaCtrlPipeline.__call__.__wrapped__.__defaults__) if hasattr(MasaCtrlPipeline.__call__, "__wrapped__") else None
    fn = MasaCtrlPipeline.__call__.__wrapped__ if d is not None else MasaCtrlPipeline.__call__
    d = list(fn.__defaults__)
    d[names.index("num_inference_steps")] = steps
    fn.__defaults__ = tuple(d)
    src, tgt = PROMPT_PAIRS[0][0], PROMPT_PAIRS[0][1]
    img_path = os.path.join(ref_shim.REF, "scripts", "example_cat.jpg")
    out = {"steps": np.int64(steps), "start_step": np.int64(start_step), "start_layer": np.int64(start_layer), "src": src, "tgt": tgt,
           "weight_seed": np.int64(seed)}
    hook = None
    if perturb is not None:     # the reference's own sensitivity: every UNet output times 1 + sigma * N(0, 1) (see null_text)
        sigma, pseed = perturb
        gen = torch.Generator().manual_seed(pseed)

        def noisy(_m, _a, o):
            e = o["sample"]
            o["sample"] = e * (1 + sigma * torch.randn(e.shape, generator=gen).to(e.dtype))
            return o

        hook = ed.model.unet.register_forward_hook(noisy)
        out.update(perturb_sigma=np.float64(sigma), perturb_seed=np.int64(pseed))
    for m in methods:
        t0 = time.time()
        decoded, stages = [], {}
        orig_l2i = ed.model.latent2image
        orig_invert = inv.DirectInversion.invert
        orig_pinv = ed.model.invert

        def l2i_spy(latents, return_type="np"):
            decoded.append(latents.clone().numpy())
            return orig_l2i(latents, return_type=return_type)

        def invert_spy(self, *a, **k):
            r = orig_invert(self, *a, **k)
            stages["x_stars"] = torch.stack([x.clone() for x in r[2]]).numpy()
            stages["noise_loss"] = torch.stack([x.clone() for x in r[3]]).numpy()
            return r

        def pinv_spy(*a, **k):
            r = orig_pinv(*a, **k)
            stages["x_stars"] = torch.stack([x.clone() for x in r[1]]).numpy()
            return r

        ed.model.latent2image = l2i_spy
        inv.DirectInversion.invert = invert_spy
        ed.model.invert = pinv_spy
        try:
            with ref_shim.cuda_to_cpu(), torch.no_grad():
                panel = ed(m, img_path, src, tgt, 7.5, step=start_step, layper=start_layer)
        finally:
            ed.model.latent2image = orig_l2i
            inv.DirectInversion.invert = orig_invert
            ed.model.invert = orig_pinv
        lat = [x for x in decoded if x.shape[0] in (1, 2) and x.ndim == 4]
        if keep_every > 1:          # the 50-step fixture: every keep_every-th inversion latent / offset and the end points
            xs_idx = sorted(set(list(range(0, steps + 1, keep_every)) + [steps]))
            nl_idx = sorted(set(list(range(0, steps, keep_every)) + [steps - 1]))
            out["x_stars_index"], out["noise_loss_index"] = np.array(xs_idx, np.int64), np.array(nl_idx, np.int64)
            stages["x_stars"] = stages["x_stars"][xs_idx]
            if "noise_loss" in stages:
                stages["noise_loss"] = stages["noise_loss"][nl_idx]
        out[m + "/x_stars"] = stages["x_stars"]
        if "noise_loss" in stages:
            out[m + "/noise_loss"] = stages["noise_loss"]
        out[m + "/fixed_latents"] = lat[-2]
        out[m + "/masactrl_latents"] = lat[-1]
        p = np.array(panel)
        out[m + "/recon_image_small"] = p[::4, 1024:1536:4]
        out[m + "/edited_image_small"] = p[::4, 1536::4]
        print("masactrl", m, "%.1fs" % (time.time() - t0), {k: v.shape for k, v in out.items() if k.startswith(m + "/")})
    if hook is not None:
        hook.remove()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def masactrl_lists(steps=6, layer_idx=(10, 12, 15), step_idx=(1, 3, 4)):
    """MutualSelfAttentionControl(layer_idx=..., step_idx=...) (models/masactrl/masactrl.py:14-39: arbitrary lists instead of the
    [start_step, total) x [start_layer, 16) windows; no shipped script passes them): the reference's MasaCtrlPipeline.__call__ on SMALL64
    from the inverted latent of e2e_masactrl.npz's ddim+masactrl run, with the reference's own controller object built by hand."""
    cfg, seed = SMALL64, 2
    usd, vsd = weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed)
    ed = ref_shim.build_masactrl_editor(cfg, usd, vsd, WordTokenizer(), SyntheticTextEncoder(cfg.cross_dim, seed=7), steps)
    from models.masactrl.masactrl import MutualSelfAttentionControl
    from models.masactrl.masactrl_utils import regiter_attention_editor_diffusers
    g = np.load(os.path.join(OUT, "e2e_masactrl.npz"))
    assert int(g["steps"]) == steps
    x_t = torch.from_numpy(g["ddim+masactrl/x_stars"][-1])
    tgt = str(g["tgt"])
    decoded = []
    orig_l2i = ed.model.latent2image

    def l2i_spy(latents, return_type="np"):
        decoded.append(latents.clone().numpy())
        return orig_l2i(latents, return_type=return_type)

    ed.model.latent2image = l2i_spy
    try:
        with ref_shim.cuda_to_cpu(), torch.no_grad():
            editor = MutualSelfAttentionControl(layer_idx=list(layer_idx), step_idx=list(step_idx), total_steps=steps)
            regiter_attention_editor_diffusers(ed.model, editor)
            ed.model(["", tgt], latents=x_t.expand(2, -1, -1, -1), num_inference_steps=steps, guidance_scale=7.5)
    finally:
        ed.model.latent2image = orig_l2i
    np.savez_compressed(os.path.join(OUT, "masactrl_lists.npz"), x_t=x_t.numpy(), latents=decoded[-1], layer_idx=np.array(layer_idx, np.int64),
                        step_idx=np.array(step_idx, np.int64), steps=np.int64(steps), tgt=tgt)
    print("masactrl_lists", decoded[-1].shape)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(int(os.environ.get("PNPI_GOLDEN_THREADS", os.cpu_count())))
    which = sys.argv[1:] or ["host", "models", "e2e", "variants", "masactrl", "proximal", "clip", "dispatch"]
    if "host" in which:
        host_tables()
    if "models" in which:
        model_goldens()
    if "e2e" in which:
        e2e("refine", False, True)
        e2e("replace", True, False)
    if "e2e_insert" in which or not sys.argv[1:]:
        # Refine with INSERTED target tokens (mapper -1 / alphas 0), with and without LocalBlend + Reweight
        e2e("insert2", False, True, pair=2)
        e2e("insert3", False, False, pair=3)
    if "e2e_sd1" in which or not sys.argv[1:]:
        # the benchmarked configuration: full SD-1.x width (859.5 M parameters), the reference's own P2PEditor, 2 + 2 steps
        e2e("sd1", False, True, steps=2, cfg=SD1, seed=0)
    if "e2e_sd1_50" in which:
        # the benchmarked configuration AND the benchmarked schedule: full SD-1.x width, 50 + 50 steps (about 45-60 min of CPU;
        # not part of the default list)
        e2e("sd1_50", False, True, steps=50, cfg=SD1, seed=0, keep_every=10)
    if "substruct" in which or not sys.argv[1:]:
        local_blend_substruct()
    if "variants" in which:
        variants()
    if "masactrl" in which:
        masactrl()
    if "proximal" in which:
        proximal()
    if "proximal_recon" in which or not sys.argv[1:]:
        proximal_recon()
    if "proximal_inv_guidance" in which or not sys.argv[1:]:
        proximal_inv_guidance()
    if "null_text" in which or not sys.argv[1:]:
        null_text()
    if "unet_ctxgrad_sd1" in which:
        unet_ctxgrad_sd1()
    if "null_text_sd1" in which:
        # BASELINE config 4 at the benchmarked width: 2 steps x 10 Adam iterations through the full-width UNet (about 10 CPU-minutes)
        null_text(steps=2, cfg=SD1, seed=0, name="e2e_null_text_sd1")
    if "masactrl_sd1" in which:
        # BASELINE config 5 at the benchmarked width: 4 steps, mutual self-attention from step 1 in transformer blocks 10..15
        masactrl(steps=4, start_step=1, start_layer=10, cfg=SD1, seed=0, name="e2e_masactrl_sd1")
    if "null_text_sd1_5" in which:
        # round 5: the same at 5 steps x 10 Adam iterations (about 25 CPU-minutes): drift of the fp16 reverse walk / Adam state over more steps
        null_text(steps=5, cfg=SD1, seed=0, name="e2e_null_text_sd1_5")
    if "null_text_sd1_20" in which:
        # round 6: 20 steps x 10 Adam iterations at full width, every 5th latent / embedding kept
        null_text(steps=20, cfg=SD1, seed=0, name="e2e_null_text_sd1_20", keep_every=5)
    if "null_text_sd1_20_pert" in which:
        # the same reference run with one fp16 rounding (relative 2^-11) on every UNet output: the reference's own sensitivity
        null_text(steps=20, cfg=SD1, seed=0, name="e2e_null_text_sd1_20_pert", keep_every=5, perturb=(2.0 ** -11, 77))
    if "null_text_sd1_5_pert" in which:
        null_text(steps=5, cfg=SD1, seed=0, name="e2e_null_text_sd1_5_pert", perturb=(2.0 ** -11, 77))
    if "e2e_sd1_replace" in which:
        # round 6: BASELINE config 2's "P2P AttentionReplace" at the full SD-1.x width: the cake pair (equal word counts),
        # is_replace_controller=True with LocalBlend + Reweight, 2 + 2 steps
        e2e("sd1_replace", True, True, steps=2, cfg=SD1, seed=0)
    if "e2e_sd1_replace_50" in which:
        # the same at the benchmarked SCHEDULE: 50 + 50 steps (about an hour of CPU), every 10th inversion latent / offset kept
        e2e("sd1_replace_50", True, True, steps=50, cfg=SD1, seed=0, keep_every=10)
    if "masactrl_sd1_10" in which:
        # round 5: 10 steps, mutual self-attention from step 3 (about 15 CPU-minutes)
        masactrl(steps=10, start_step=3, start_layer=10, cfg=SD1, seed=0, name="e2e_masactrl_sd1_10")
    if "masactrl_sd1_50" in which:
        # round 6: BASELINE config 5 at the benchmarked schedule -- 50 steps, the editor's defaults (mutual self-attention from step 4 in blocks 10..15)
        masactrl(steps=50, start_step=4, start_layer=10, cfg=SD1, seed=0, name="e2e_masactrl_sd1_50", keep_every=10)
    if "masactrl_sd1_50_pert" in which:
        # the reference's own sensitivity of the uncorrected path (ddim+masactrl: plain DDIM sampling at guidance 7.5 from the inverted latent --
        # the divergence direct inversion exists to remove): the same run with one fp16 rounding on every UNet output
        masactrl(steps=50, start_step=4, start_layer=10, cfg=SD1, seed=0, name="e2e_masactrl_sd1_50_pert", keep_every=10, perturb=(2.0 ** -11, 77),
                 methods=("ddim+masactrl",))
    if "masactrl_lists" in which:
        masactrl_lists()
    if "null_latent" in which or not sys.argv[1:]:
        null_latent()
    if "null_text_family" in which or not sys.argv[1:]:
        null_text_family()
    if "clip" in which:
        clip_text()
    if "dispatch" in which:
        dispatch()

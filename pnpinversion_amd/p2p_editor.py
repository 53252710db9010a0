"""P2PEditor with the constructor / call signature of the reference's models/p2p_editor.py:12-45, on the native pipeline.

Implemented method strings (Appendix D of SURVEY.md; models/p2p_editor.py:46-135): "directinversion+p2p" (the hot path) and
every loop variant of it that needs no new kernel -- "ddim+p2p", "negative-prompt-inversion+p2p", the 20
"directinversion+p2p_guidance_<inv>_<fwd>" strings, "ablation_directinversion_{04,08}+p2p",
"ablation_directinversion_interval_{2,5,10,24,49}+p2p", "ablation_directinversion_add-target+p2p" / "...add-source+p2p".
"null-text-inversion+p2p" (and its two aliases), "ablation_null-text-inversion_single_branch+p2p" and
"null-text-inversion+proximal-guidance" run the device null-text optimisation (pnpi_null_text_optimize),
"ablation_null-latent-inversion+p2p" its latent-offset variant (pnpi_null_latent_calculate); any other string raises the reference's
NotImplementedError(f"No edit method named {edit_method}") (models/p2p_editor.py:134-135)."""
import numpy as np
from PIL import Image

from .config import SD1
from .p2p.attention_control import AttentionStore, make_controller
from .p2p.inversion import DirectInversion, NegativePromptInversion, NullInversion
from .p2p.p2p_guidance_forward import (direct_inversion_p2p_guidance_forward,
                                       direct_inversion_p2p_guidance_forward_add_target, p2p_guidance_forward,
                                       p2p_guidance_forward_single_branch)
from .p2p.proximal_guidance_forward import proximal_guidance_forward
from .p2p.attention_control import register_attention_control
from .pipeline import NativePipeline
from .utils.utils import image2latent, latent2image, load_512, txt_draw
import torch


class P2PEditor:
    def __init__(self, method_list, device, num_ddim_steps=50, *, pipeline=None, cfg=SD1, weight_seed=0, state_dicts=None,
                 tokenizer=None, checkpoint_dir=None):
        self.device = device
        self.method_list = method_list
        self.num_ddim_steps = num_ddim_steps
        if pipeline is None:
            # The reference loads CompVis/stable-diffusion-v1-4 here (models/p2p_editor.py:23-24): `checkpoint_dir` is that
            # directory in the diffusers layout (unet/, vae/, text_encoder/, tokenizer/ -> checkpoint.load_checkpoint_dir).
            # Alternatively state_dicts = (unet, vae[, clip_text_model]) with `tokenizer` = the checkpoint's CLIP tokenizer
            # (text.ClipBPETokenizer(vocab.json, merges.txt) or transformers' CLIPTokenizer).  With neither, seeded synthetic weights
            # and the word-level stand-in tokenizer (no checkpoint exists on the build / GPU boxes).
            if checkpoint_dir is not None:
                from .checkpoint import load_checkpoint_dir
                unet_sd, vae_sd, clip_sd, ck_tok = load_checkpoint_dir(checkpoint_dir)
                state_dicts, tokenizer = (unet_sd, vae_sd, clip_sd), tokenizer or ck_tok
            if state_dicts is not None:
                native_text = len(state_dicts) > 2 and state_dicts[2] is not None
                if native_text and tokenizer is None:
                    # real CLIPTextModel weights fed with the stand-in tokenizer's hashed ids would embed noise: a silently wrong edit
                    raise ValueError("CLIP text-encoder weights were supplied without their tokenizer: pass tokenizer= (e.g. "
                                     "pnpinversion_amd.text.ClipBPETokenizer(vocab.json, merges.txt)) or checkpoint_dir=")
                pipeline = NativePipeline(cfg, device=device, text_encoder="native" if native_text else None, tokenizer=tokenizer)
                pipeline.load_state_dict(state_dicts[0], state_dicts[1], clip_sd=state_dicts[2] if native_text else None)
            else:
                pipeline = NativePipeline.synthetic(cfg, seed=weight_seed, device=device, text_encoder="native", tokenizer=tokenizer)
        self.ldm_stable = pipeline
        self.scheduler = pipeline.scheduler
        # lock-step schedule (pnpi_direct_edit): offsets + reconstruction pass + edit pass share one UNet launch per timestep.
        # False runs the reference's phase order call by call (invert -> forward -> forward); same results within tolerance.
        self.lockstep = True
        # "faithful" (default): every UNet call of the reference is executed (650 sample-forwards per directinversion+p2p image).
        # "pruned": the algebraically equivalent schedule of SURVEY.md Note D for "directinversion+p2p" -- the source latent of the
        # edit pass is assigned from the stored inversion trajectory (it equals prev + offset exactly), which makes the offset pass,
        # the reconstruction pass and the unconditional-source row redundant: 200 sample-forwards.  Opt-in; same panels.
        self.schedule = "faithful"
        self.ldm_stable.scheduler.set_timesteps(self.num_ddim_steps)

    def __call__(self, edit_method, image_path, prompt_src, prompt_tar, guidance_scale=7.5, proximal=None, quantile=0.7,
                 use_reconstruction_guidance=False, recon_t=400, recon_lr=0.1, cross_replace_steps=0.4, self_replace_steps=0.6,
                 blend_word=None, eq_params=None, is_replace_controller=False, use_inversion_guidance=False, dilate_mask=1):
        kw = dict(guidance_scale=guidance_scale, cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps,
                  blend_word=blend_word, eq_params=eq_params, is_replace_controller=is_replace_controller)
        if edit_method == "directinversion+p2p":
            return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, **kw)
        if edit_method == "ddim+p2p":
            return self.edit_image_ddim(image_path, prompt_src, prompt_tar, **kw)
        if edit_method == "negative-prompt-inversion+p2p":
            return self.edit_image_negative_prompt_inversion(image_path, prompt_src, prompt_tar, proximal=None, **kw)
        if edit_method.startswith("directinversion+p2p_guidance_"):
            table = {"0": 0, "1": 1, "25": 2.5, "5": 5, "75": 7.5}          # models/p2p_editor.py:78-88
            parts = edit_method.split("_")
            if len(parts) == 4 and parts[-2] in table and parts[-1] in ("1", "5", "25", "75"):
                kw.pop("guidance_scale")
                return self.edit_image_directinversion_vary_guidance_scale(image_path, prompt_src, prompt_tar,
                                                                           inverse_guidance_scale=table[parts[-2]],
                                                                           forward_guidance_scale=table[parts[-1]], **kw)
        if edit_method in ("ablation_directinversion_08+p2p", "ablation_directinversion_04+p2p"):
            scale = float(edit_method.split("+")[0].split("_")[-1]) / 10
            return self.edit_image_directinversion_not_full(image_path, prompt_src, prompt_tar, scale=scale, **kw)
        if edit_method in ("ablation_directinversion_interval_2+p2p", "ablation_directinversion_interval_5+p2p",
                           "ablation_directinversion_interval_10+p2p", "ablation_directinversion_interval_24+p2p",
                           "ablation_directinversion_interval_49+p2p"):
            skip_step = int(edit_method.split("+")[0].split("_")[-1])
            return self.edit_image_directinversion_skip_step(image_path, prompt_src, prompt_tar, skip_step=skip_step, **kw)
        if edit_method == "ablation_directinversion_add-target+p2p":
            return self.edit_image_directinversion_add_target(image_path, prompt_src, prompt_tar, **kw)
        if edit_method == "ablation_directinversion_add-source+p2p":
            return self.edit_image_directinversion_add_source(image_path, prompt_src, prompt_tar, **kw)
        if edit_method in ("null-text-inversion+p2p", "null-text-inversion+p2p_a800", "null-text-inversion+p2p_3090"):
            return self.edit_image_null_text_inversion(image_path, prompt_src, prompt_tar, **kw)
        if edit_method == "ablation_null-text-inversion_single_branch+p2p":
            return self.edit_image_null_text_inversion_single_branch(image_path, prompt_src, prompt_tar, **kw)
        if edit_method == "null-text-inversion+proximal-guidance":
            return self.edit_image_null_text_inversion_proximal_guidanca(image_path, prompt_src, prompt_tar, proximal=proximal, quantile=quantile,
                                                                         use_reconstruction_guidance=use_reconstruction_guidance,
                                                                         recon_t=recon_t, recon_lr=recon_lr,
                                                                         use_inversion_guidance=use_inversion_guidance, dilate_mask=dilate_mask, **kw)
        if edit_method == "ablation_null-latent-inversion+p2p":
            return self.edit_image_null_latent_inversion(image_path, prompt_src, prompt_tar, **kw)
        if edit_method == "negative-prompt-inversion+proximal-guidance":
            return self.edit_image_negative_prompt_inversion(image_path, prompt_src, prompt_tar, proximal=proximal, quantile=quantile,
                                                             use_reconstruction_guidance=use_reconstruction_guidance,
                                                             recon_t=recon_t, recon_lr=recon_lr,
                                                             use_inversion_guidance=use_inversion_guidance, dilate_mask=dilate_mask, **kw)
        raise NotImplementedError(f"No edit method named {edit_method}")

    def edit_image_directinversion(self, image_path, prompt_src, prompt_tar, guidance_scale=7.5, cross_replace_steps=0.4,
                                   self_replace_steps=0.6, blend_word=None, eq_params=None, is_replace_controller=False,
                                   add_target=False, return_stages=False, inverse_guidance_scale=None, offset_scale=None,
                                   null_latent=False):
        """models/p2p_editor.py:415-479; with the keyword-only knobs also :481-548 (vary guidance: CFG inversion at
        inverse_guidance_scale, everything else at guidance_scale), :707-773 (not_full: offset_scale = scale) and :775-840
        (skip_step: offset_scale = per-step 0/1 list)."""
        forward = direct_inversion_p2p_guidance_forward_add_target if add_target else direct_inversion_p2p_guidance_forward
        image_gt = load_512(image_path)
        side = self.ldm_stable.engine.cfg.sample_size * self.ldm_stable.engine.cfg.vae_scale
        if side != 512:   # reduced test configurations only; the reference is fixed at 512 (utils/utils.py:45)
            image_gt = np.array(Image.fromarray(image_gt).resize((side, side)))
        prompts = [prompt_src, prompt_tar]
        null_inversion = DirectInversion(model=self.ldm_stable, num_ddim_steps=self.num_ddim_steps)
        if self.schedule not in ("faithful", "pruned"):
            raise ValueError("P2PEditor.schedule must be 'faithful' or 'pruned'")
        if null_latent:
            # models/p2p_editor.py:640-705: the offsets come from DirectInversion.invert_null_latent (an optimisation through the UNet per
            # step), the two guidance passes are the direct-inversion ones; phase order of the reference (nothing to run in lock step)
            _, _, x_stars, noise_loss_list = null_inversion.invert_null_latent(image_gt=image_gt, prompt=prompts, guidance_scale=guidance_scale)
        elif (self.lockstep and self.ldm_stable.engine.max_unet_rows >= 12) or self.schedule == "pruned":
            return self._edit_lockstep(null_inversion, image_gt, prompts, prompt_src, prompt_tar, guidance_scale, cross_replace_steps,
                                       self_replace_steps, blend_word, eq_params, is_replace_controller, add_target, return_stages, side,
                                       inverse_guidance_scale, offset_scale)
        if not null_latent:
            null_inversion.init_prompt(prompts)
            register_attention_control(self.ldm_stable, None)
            if inverse_guidance_scale is None:
                _, x_stars = null_inversion.ddim_inversion(image_gt)
            else:
                _, x_stars = null_inversion.ddim_with_guidance_scale_inversion(image_gt, inverse_guidance_scale)
            noise_loss_list = null_inversion._offsets(x_stars, guidance_scale, offset_scale)
        x_t = x_stars[-1]
        controller = AttentionStore()
        reconstruct_latent, x_t = forward(model=self.ldm_stable, prompt=prompts, controller=controller, noise_loss_list=noise_loss_list,
                                          latent=x_t, num_inference_steps=self.num_ddim_steps, guidance_scale=guidance_scale,
                                          generator=None)
        reconstruct_image = latent2image(model=self.ldm_stable.vae, latents=reconstruct_latent)[0]
        cross_replace_steps = {"default_": cross_replace_steps}
        controller = make_controller(pipeline=self.ldm_stable, prompts=prompts, is_replace_controller=is_replace_controller,
                                     cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps,
                                     blend_words=blend_word, equilizer_params=eq_params, num_ddim_steps=self.num_ddim_steps,
                                     device=self.device)
        latents, _ = forward(model=self.ldm_stable, prompt=prompts, controller=controller, noise_loss_list=noise_loss_list,
                             latent=x_t, num_inference_steps=self.num_ddim_steps, guidance_scale=guidance_scale, generator=None)
        images = latent2image(model=self.ldm_stable.vae, latents=latents)
        image_instruct = txt_draw(f"source prompt: {prompt_src}\ntarget prompt: {prompt_tar}", target_size=(side, side))
        panel = Image.fromarray(np.concatenate((image_instruct, image_gt, reconstruct_image, images[-1]), axis=1))
        if return_stages:
            return panel, dict(x_stars=x_stars, noise_loss_list=noise_loss_list, reconstruct_latent=reconstruct_latent,
                               latents=latents, reconstruct_image=reconstruct_image, edited_image=images[-1])
        return panel

    def edit_image_directinversion_vary_guidance_scale(self, image_path, prompt_src, prompt_tar, inverse_guidance_scale=1,
                                                       forward_guidance_scale=7.5, **kw):
        """models/p2p_editor.py:481-548"""
        return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, guidance_scale=forward_guidance_scale,
                                               inverse_guidance_scale=inverse_guidance_scale, **kw)

    def edit_image_directinversion_not_full(self, image_path, prompt_src, prompt_tar, guidance_scale=7.5, scale=1., **kw):
        """models/p2p_editor.py:707-773"""
        return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, guidance_scale=guidance_scale,
                                               offset_scale=float(scale), **kw)

    def edit_image_directinversion_skip_step(self, image_path, prompt_src, prompt_tar, skip_step, guidance_scale=7.5, **kw):
        """models/p2p_editor.py:775-840"""
        sc = [1.0 if i % skip_step == 0 else 0.0 for i in range(self.num_ddim_steps)]
        return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, guidance_scale=guidance_scale, offset_scale=sc, **kw)

    def edit_image_directinversion_add_target(self, image_path, prompt_src, prompt_tar, **kw):
        """models/p2p_editor.py:842-907: the offset is added to both branches (p2p_guidance_forward.py:119-132)"""
        return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, add_target=True, **kw)

    def edit_image_directinversion_add_source(self, image_path, prompt_src, prompt_tar, **kw):
        """models/p2p_editor.py:909-978: the same code path as add_target in the reference"""
        return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, add_target=True, **kw)

    def _plain_p2p(self, forward, image_gt, x_stars, uncond_embeddings, prompt_src, prompt_tar, guidance_scale, cross_replace_steps,
                   self_replace_steps, blend_word, eq_params, is_replace_controller, side, return_stages):
        """Shared tail of edit_image_ddim / edit_image_negative_prompt_inversion (p2p_editor.py:158-196, 351-411): an
        AttentionStore reconstruction of the source prompt alone, then the controlled pair, both without offsets."""
        x_t = x_stars[-1]
        reconstruct_latent, x_t = forward(model=self.ldm_stable, prompt=[prompt_src], controller=AttentionStore(), latent=x_t,
                                          guidance_scale=guidance_scale, generator=None, uncond_embeddings=uncond_embeddings)
        reconstruct_image = latent2image(model=self.ldm_stable.vae, latents=reconstruct_latent)[0]
        controller = make_controller(pipeline=self.ldm_stable, prompts=[prompt_src, prompt_tar],
                                     is_replace_controller=is_replace_controller,
                                     cross_replace_steps={"default_": cross_replace_steps}, self_replace_steps=self_replace_steps,
                                     blend_words=blend_word, equilizer_params=eq_params, num_ddim_steps=self.num_ddim_steps,
                                     device=self.device)
        latents, _ = forward(model=self.ldm_stable, prompt=[prompt_src, prompt_tar], controller=controller, latent=x_t,
                             guidance_scale=guidance_scale, generator=None, uncond_embeddings=uncond_embeddings)
        images = latent2image(model=self.ldm_stable.vae, latents=latents)
        image_instruct = txt_draw(f"source prompt: {prompt_src}\ntarget prompt: {prompt_tar}", target_size=(side, side))
        panel = Image.fromarray(np.concatenate((image_instruct, image_gt, reconstruct_image, images[-1]), axis=1))
        if return_stages:
            return panel, dict(x_stars=x_stars, reconstruct_latent=reconstruct_latent, latents=latents)
        return panel

    def _load(self, image_path):
        image_gt = load_512(image_path)
        side = self.ldm_stable.engine.cfg.sample_size * self.ldm_stable.engine.cfg.vae_scale
        if side != 512:   # reduced test configurations only
            image_gt = np.array(Image.fromarray(image_gt).resize((side, side)))
        return image_gt, side

    def edit_image_ddim(self, image_path, prompt_src, prompt_tar, guidance_scale=7.5, cross_replace_steps=0.4,
                        self_replace_steps=0.6, blend_word=None, eq_params=None, is_replace_controller=False, return_stages=False):
        """models/p2p_editor.py:137-197: DDIM inversion of the source prompt, then plain Prompt-to-Prompt (no correction)."""
        image_gt, side = self._load(image_path)
        self.ldm_stable.scheduler.set_timesteps(self.num_ddim_steps)
        inv = NullInversion(model=self.ldm_stable, num_ddim_steps=self.num_ddim_steps)
        _, _, x_stars, uncond_embeddings = inv.invert(image_gt=image_gt, prompt=prompt_src, guidance_scale=guidance_scale,
                                                      num_inner_steps=0)
        fwd = lambda **k: p2p_guidance_forward(num_inference_steps=self.num_ddim_steps, **k)   # noqa: E731
        return self._plain_p2p(fwd, image_gt, x_stars, uncond_embeddings, prompt_src, prompt_tar, guidance_scale, cross_replace_steps,
                               self_replace_steps, blend_word, eq_params, is_replace_controller, side, return_stages)

    def edit_image_null_text_inversion(self, image_path, prompt_src, prompt_tar, guidance_scale=7.5, cross_replace_steps=0.4,
                                       self_replace_steps=0.6, blend_word=None, eq_params=None, is_replace_controller=False,
                                       single_branch=False, proximal=None, quantile=0.7, use_reconstruction_guidance=False,
                                       recon_t=400, recon_lr=0.1, use_inversion_guidance=False, dilate_mask=1,
                                       num_inner_steps=10, return_stages=False):
        """models/p2p_editor.py:199-259 (null-text inversion + P2P), :261-322 (single branch), :550-638 (+ proximal guidance, with the
        reconstruction guidance of :620-627 when use_reconstruction_guidance is set): NullInversion.invert = DDIM inversion + per-step
        optimisation of the unconditional embedding (pnpi_null_text_optimize), then the two guidance passes with the per-step embeddings."""
        image_gt, side = self._load(image_path)
        self.ldm_stable.scheduler.set_timesteps(self.num_ddim_steps)
        inv = NullInversion(model=self.ldm_stable, num_ddim_steps=self.num_ddim_steps)
        _, _, x_stars, uncond_embeddings = inv.invert(image_gt=image_gt, prompt=prompt_src, guidance_scale=guidance_scale,
                                                      num_inner_steps=num_inner_steps)
        # Reconstruction guidance pulls the predicted x0 towards the ENCODED source image.  The reference hands this editor
        # NullInversion.invert's second return value for it -- the decoded uint8 image (inversion.py:190-194,227-234), which
        # DDIMSchedulerDev.step cannot subtract from a latent: its use_reconstruction_guidance=True raises.  The encoded latent is
        # x*_0 = ddim_latents[0], what NegativePromptInversion.invert returns in that position (inversion.py:72-76,96) for the same code.
        image_enc_latent = x_stars[0]
        base = p2p_guidance_forward_single_branch if single_branch else p2p_guidance_forward
        on = use_reconstruction_guidance or use_inversion_guidance

        def fwd(**k):   # the reconstruction pass runs without the proximal step (edit_stage=False, p2p_editor.py:577-594)
            if proximal is not None and len(k["prompt"]) == 2:
                return proximal_guidance_forward(num_inference_steps=self.num_ddim_steps, edit_stage=True, prox=proximal, quantile=quantile,
                                                 image_enc=image_enc_latent if use_reconstruction_guidance else None,
                                                 recon_lr=recon_lr if on else 0, recon_t=recon_t if on else 1000, dilate_mask=dilate_mask,
                                                 inversion_guidance=False, x_stars=x_stars, **k)     # as p2p_editor.py:593,632
            return base(num_inference_steps=self.num_ddim_steps, **k)
        out = self._plain_p2p(fwd, image_gt, x_stars, uncond_embeddings, prompt_src, prompt_tar, guidance_scale, cross_replace_steps,
                              self_replace_steps, blend_word, eq_params, is_replace_controller, side, return_stages)
        if return_stages:
            out[1]["uncond_embeddings"] = uncond_embeddings
            out[1]["inner_losses"] = getattr(inv, "inner_losses", None)
        return out

    def edit_image_null_text_inversion_single_branch(self, image_path, prompt_src, prompt_tar, **kw):
        """models/p2p_editor.py:261-322"""
        return self.edit_image_null_text_inversion(image_path, prompt_src, prompt_tar, single_branch=True, **kw)

    def edit_image_null_text_inversion_proximal_guidanca(self, image_path, prompt_src, prompt_tar, proximal=None, quantile=0.7,
                                                         use_reconstruction_guidance=False, recon_t=400, recon_lr=0.1,
                                                         use_inversion_guidance=False, dilate_mask=1, **kw):
        """models/p2p_editor.py:550-638 (the reference's spelling).  recon_* / use_inversion_guidance / dilate_mask only matter with
        reconstruction guidance (image_enc is None otherwise, and the inversion-guidance branch is dead code: :87-89 precedence)."""
        return self.edit_image_null_text_inversion(image_path, prompt_src, prompt_tar, proximal=proximal, quantile=quantile,
                                                   use_reconstruction_guidance=use_reconstruction_guidance, recon_t=recon_t, recon_lr=recon_lr,
                                                   use_inversion_guidance=use_inversion_guidance, dilate_mask=dilate_mask, **kw)

    def edit_image_null_latent_inversion(self, image_path, prompt_src, prompt_tar, **kw):
        """models/p2p_editor.py:640-705: DirectInversion.invert_null_latent (pnpi_null_latent_calculate) + the two direct-inversion passes"""
        return self.edit_image_directinversion(image_path, prompt_src, prompt_tar, null_latent=True, **kw)

    def edit_image_negative_prompt_inversion(self, image_path, prompt_src, prompt_tar, guidance_scale=7.5, proximal=None,
                                             quantile=0.7, use_reconstruction_guidance=False, recon_t=400, recon_lr=0.1, npi_interp=0,
                                             cross_replace_steps=0.4, self_replace_steps=0.6, blend_word=None, eq_params=None,
                                             is_replace_controller=False, use_inversion_guidance=False, dilate_mask=1,
                                             return_stages=False):
        """models/p2p_editor.py:324-413: the source prompt's embedding replaces "" in the unconditional rows; with proximal
        "l0" / "l1" the edit pass soft-thresholds the CFG difference (proximal_guidance_forward.py:39-64)."""
        image_gt, side = self._load(image_path)
        self.ldm_stable.scheduler.set_timesteps(self.num_ddim_steps)
        inv = NegativePromptInversion(model=self.ldm_stable, num_ddim_steps=self.num_ddim_steps)
        _, image_enc_latent, x_stars, uncond_embeddings = inv.invert(image_gt=image_gt, prompt=prompt_src, npi_interp=npi_interp)
        on = use_reconstruction_guidance or use_inversion_guidance
        def fwd(**k):   # reconstruction: edit_stage=False (no proximal step); edit: the method's prox / quantile (p2p_editor.py:389-405)
            edit = len(k["prompt"]) == 2
            return proximal_guidance_forward(edit_stage=edit, prox=proximal if edit else None, quantile=quantile,
                                             image_enc=image_enc_latent if (edit and use_reconstruction_guidance) else None,
                                             recon_lr=recon_lr if on else 0, recon_t=recon_t if on else 1000, dilate_mask=dilate_mask,
                                             inversion_guidance=False, x_stars=x_stars if edit else None,       # as p2p_editor.py:368-369,407
                                             num_inference_steps=self.num_ddim_steps, **k)
        return self._plain_p2p(fwd, image_gt, x_stars, uncond_embeddings, prompt_src, prompt_tar, guidance_scale, cross_replace_steps,
                               self_replace_steps, blend_word, eq_params, is_replace_controller, side, return_stages)

    @torch.no_grad()
    def _edit_lockstep(self, inv, image_gt, prompts, prompt_src, prompt_tar, guidance_scale, cross_replace_steps, self_replace_steps,
                       blend_word, eq_params, is_replace_controller, add_target, return_stages, side, inverse_guidance_scale=None,
                       offset_scale=None):
        """Same phases as models/p2p_editor.py:415-479, re-scheduled: after the 50 B=1 inversion steps, offset_calculate
        (inversion.py:375-391), the AttentionStore reconstruction pass and the edit pass (p2p_guidance_forward.py:135-173) walk
        the same 50 timesteps and only exchange noise_loss[i] at step i, so they run as ONE 12-row UNet launch per step."""
        model = self.ldm_stable
        x_stars = self._invert_stage(model, inv, image_gt, prompts, inverse_guidance_scale)
        return self._edit_stage(x_stars, inv.context, image_gt, prompts, prompt_src, prompt_tar, guidance_scale, cross_replace_steps,
                                self_replace_steps, blend_word, eq_params, is_replace_controller, add_target, return_stages, side, offset_scale)

    def _invert_stage(self, model, inv, image_gt, prompts, inverse_guidance_scale=None):
        """Stage 1 of an edit on `model`: prompt embedding, VAE encode (+ the reference's discarded decode), the 50 one-row inversion steps."""
        model.scheduler.set_timesteps(self.num_ddim_steps)
        inv.init_prompt(prompts)
        register_attention_control(model, None)
        if inverse_guidance_scale is None:
            _, x_stars = inv.ddim_inversion(image_gt)
        else:
            _, x_stars = inv.ddim_with_guidance_scale_inversion(image_gt, inverse_guidance_scale)
        return x_stars

    def _edit_stage(self, x_stars, context, image_gt, prompts, prompt_src, prompt_tar, guidance_scale, cross_replace_steps, self_replace_steps,
                    blend_word, eq_params, is_replace_controller, add_target, return_stages, side, offset_scale=None):
        """Stage 2 on the main pipeline: the lock-step (or pruned) loop from the inversion trajectory, the decodes, the panel."""
        model = self.ldm_stable
        model.scheduler.set_timesteps(self.num_ddim_steps)
        controller = make_controller(pipeline=model, prompts=prompts, is_replace_controller=is_replace_controller,
                                     cross_replace_steps={"default_": cross_replace_steps}, self_replace_steps=self_replace_steps,
                                     blend_words=blend_word, equilizer_params=eq_params, num_ddim_steps=self.num_ddim_steps,
                                     device=self.device)
        register_attention_control(model, controller)
        if self.schedule == "pruned":
            if add_target or offset_scale is not None:
                raise NotImplementedError("the pruned schedule is the plain directinversion+p2p edit (full offset on the source row only)")
            out = model.engine.direct_edit_pruned(torch.stack(list(x_stars)), context[None], [controller.tables()],
                                                  model.scheduler.timesteps.numpy(), guidance_scale)
            controller.cur_step += self.num_ddim_steps
            image_instruct = txt_draw(f"source prompt: {prompt_src}\ntarget prompt: {prompt_tar}", target_size=(side, side))
            noise_loss_list = None                                     # never materialised: latents[0] := x*_{t-1}
            latents = out[0]
            reconstruct_latent = torch.stack([out[0, 0], out[0, 0]])   # the reconstruction pass only reproduces x*_0 (Note D i)
        else:
            nl, lats = model.engine.direct_edit(torch.stack(list(x_stars)), context[None], [None, [controller.tables()]],
                                                model.scheduler.timesteps.numpy(), guidance_scale, offset_rows=2 if add_target else 1,
                                                offset_scale=offset_scale)
            controller.cur_step += self.num_ddim_steps
            # host-side panel work while the device is still in the loop (the calls above only enqueue)
            image_instruct = txt_draw(f"source prompt: {prompt_src}\ntarget prompt: {prompt_tar}", target_size=(side, side))
            noise_loss_list = [nl[i, 0] for i in range(nl.shape[0])]
            reconstruct_latent, latents = lats[0, 0], lats[1, 0]
        reconstruct_image = latent2image(model=model.vae, latents=reconstruct_latent)[0]
        images = latent2image(model=model.vae, latents=latents)
        panel = Image.fromarray(np.concatenate((image_instruct, image_gt, reconstruct_image, images[-1]), axis=1))
        if return_stages:
            return panel, dict(x_stars=x_stars, noise_loss_list=noise_loss_list, reconstruct_latent=reconstruct_latent,
                               latents=latents, reconstruct_image=reconstruct_image, edited_image=images[-1])
        return panel

    def _second_pipeline(self, rows=4):
        """A second library context on its own HIP stream, borrowing the main context's packed weight arena (pnpi_create_shared): the inversion of
        the NEXT image (rows = 4) or of the next BATCH of images (rows = images per batch) runs there while this one's lock-step loop runs
        on the main context."""
        inverter = getattr(self, "_inverter", None)
        if inverter is not None and inverter.engine.max_unet_rows < rows:       # a wider batch than the context was built for: rebuild it
            torch.cuda.synchronize(self.ldm_stable.device)
            inverter.engine.close()
            inverter = self._inverter = None
        if inverter is None:
            main = self.ldm_stable
            torch.cuda.synchronize(main.device)
            self._inv_stream = torch.cuda.Stream(device=main.device)
            with torch.cuda.device(main.device), torch.cuda.stream(self._inv_stream):
                p = main.peer(max_unet_rows=max(4, rows), max_vae_images=2)
                self._inv_stream.synchronize()
                p.scheduler.set_timesteps(self.num_ddim_steps)
            self._inverter = p
        return self._inverter

    def _peer_editors(self, n):
        """n further P2PEditors, each on a library context of its own (own HIP stream, own workspaces, the SAME packed weight arena --
        pnpi_create_shared borrows the main context's, no copy): `edit_stream_in_flight` spreads the images of a sweep over this editor and them."""
        peers = self.__dict__.setdefault("_peers", [])
        main = self.ldm_stable
        while len(peers) < n:
            torch.cuda.synchronize(main.device)
            stream = torch.cuda.Stream(device=main.device)
            with torch.cuda.device(main.device), torch.cuda.stream(stream):
                # one image per context: 12 UNet rows (the lock-step loop) are the most any method string launches
                p = main.peer(max_unet_rows=min(main.engine.max_unet_rows, 12), max_vae_images=min(main.engine.max_vae_images, 2))
                stream.synchronize()
            peer = P2PEditor(self.method_list, self.device, num_ddim_steps=self.num_ddim_steps, pipeline=p)
            peers.append((peer, stream))
        for peer, _ in peers:
            peer.lockstep, peer.schedule = self.lockstep, self.schedule
        return peers[:n]

    def close_peers(self):
        """Free the extra library contexts `edit_stream_in_flight` and `edit_stream_directinversion` created (they borrow the main
        context's weight arena: close them before the main pipeline's engine)."""
        for peer, _ in self.__dict__.pop("_peers", []):
            peer.ldm_stable.engine.close()
        inverter = self.__dict__.pop("_inverter", None)
        if inverter is not None:
            inverter.engine.close()

    def edit_stream_in_flight(self, edit_method, items, n_flight=2, **kw):
        """ANY method string over a sequence of images with n_flight images in flight: image i runs on context i % n_flight (this
        editor's, or one of n_flight - 1 further contexts with their own HIP streams), each context fed by its own worker thread.  The
        one-row launch chains of an edit (DDIM inversion; every forward and reverse walk of the null-text optimisation) keep a fraction of
        the chip busy between dependent launches -- independent chains fill the gaps.  Same kernels on the same inputs as
        `editor(edit_method, ...)` image by image -> identical panels; only sweep throughput changes (the reference's sweep visits images
        one by one, run_editing_p2p.py:239-300).  items: iterable of (image_path | array, prompt_src, prompt_tar[, blend_word[,
        eq_params]]); kw: the other keyword arguments of __call__.  Generator of panels, in order."""
        from concurrent.futures import ThreadPoolExecutor
        items = list(items)
        if n_flight < 1:
            raise ValueError("n_flight must be >= 1")
        lanes = [(self, None)] + self._peer_editors(n_flight - 1)
        dev = self.ldm_stable.device

        def run(editor, stream, it):
            extra = {}
            if len(it) > 3:
                extra["blend_word"] = it[3]
            if len(it) > 4:
                extra["eq_params"] = it[4]
            with torch.no_grad(), torch.cuda.device(dev):
                if stream is None:
                    return editor(edit_method, it[0], it[1], it[2], **extra, **kw)
                with torch.cuda.stream(stream):
                    out = editor(edit_method, it[0], it[1], it[2], **extra, **kw)
                    stream.synchronize()
                    return out

        # One single-worker queue per context (never two edits on one context), fed through a BOUNDED window: at most 2 * n_flight images
        # are submitted ahead of the in-order consumer, each future is dropped once its panel has been handed over, and a failing image
        # (or a consumer that stops early) cancels what has not started -- the reference's sweep also stops at the failing image.
        from collections import deque
        pools = [ThreadPoolExecutor(max_workers=1) for _ in lanes]
        window, nxt = deque(), 0

        def top_up():
            nonlocal nxt
            while nxt < len(items) and len(window) < 2 * n_flight:
                lane = nxt % n_flight
                window.append(pools[lane].submit(run, lanes[lane][0], lanes[lane][1], items[nxt]))
                nxt += 1

        try:
            top_up()
            while window:
                out = window.popleft().result()
                top_up()
                yield out
        finally:
            window.clear()
            for pool in pools:
                pool.shutdown(wait=True, cancel_futures=True)

    def edit_stream_two_in_flight(self, edit_method, items, **kw):
        return self.edit_stream_in_flight(edit_method, items, n_flight=2, **kw)

    def edit_stream_directinversion(self, items, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
                                    is_replace_controller=False):
        """`directinversion+p2p` over a sequence of images with stage overlap: while image i runs its 50 twelve-row lock-step steps on
        the main context, image i+1's prompt embedding, VAE encode and 50 ONE-row inversion steps (launches that fill a fraction of the
        chip) run on a second context / HIP stream from a worker thread.  Same kernels on the same inputs as `edit_image_directinversion`
        -> identical panels (tests/test_gpu_loops.py); only sweep throughput changes -- the PIE-Bench loop of run_editing_p2p.py:239-300
        visits images one by one.  items: iterable of (image_path | array, prompt_src, prompt_tar, blend_word, eq_params).  Generator of panels."""
        from concurrent.futures import ThreadPoolExecutor
        if self.schedule not in ("faithful", "pruned"):
            raise ValueError("P2PEditor.schedule must be 'faithful' or 'pruned'")
        items = list(items)
        main = self.ldm_stable
        inverter = self._second_pipeline()

        def stage1(it):
            with torch.no_grad(), torch.cuda.device(main.device), torch.cuda.stream(self._inv_stream):
                image_gt, side = self._load(it[0])
                inv = DirectInversion(model=inverter, num_ddim_steps=self.num_ddim_steps)
                x = torch.stack(list(self._invert_stage(inverter, inv, image_gt, [it[1], it[2]])))
                ctx = inv.context.clone()
                self._inv_stream.synchronize()          # the consumer is another stream: hand over finished tensors
            return image_gt, side, x, ctx

        with ThreadPoolExecutor(max_workers=1) as ex, torch.no_grad():
            fut = ex.submit(stage1, items[0]) if items else None
            for i, it in enumerate(items):
                image_gt, side, x, ctx = fut.result()
                for t in (x, ctx):
                    if t.is_cuda:                        # produced on the worker's stream, consumed on this one
                        t.record_stream(torch.cuda.current_stream(main.device))
                if i + 1 < len(items):
                    fut = ex.submit(stage1, items[i + 1])
                yield self._edit_stage(x, ctx, image_gt, [it[1], it[2]], it[1], it[2], guidance_scale, cross_replace_steps, self_replace_steps,
                                       it[3] if len(it) > 3 else None, it[4] if len(it) > 4 else None, is_replace_controller, False, False, side)

    @torch.no_grad()
    def edit_images_directinversion(self, image_paths, prompts_src, prompts_tar, guidance_scale=7.5, cross_replace_steps=0.4,
                                    self_replace_steps=0.6, blend_words=None, eq_params=None, is_replace_controller=False,
                                    add_target=False, return_stages=False):
        """`edit_image_directinversion` for a batch of images in one set of launches (the PIE-Bench sweep of
        run_editing_p2p.py:239-300 visits images one by one; one GPU has room for many): the n inversions run as n-row
        launches, the three 4-row passes of every image as one 12n-row launch per timestep.  Images are independent rows in
        every kernel; per-image results equal the single-image call within the fp16 tolerance (tests/test_gpu_loops.py).
        blend_words / eq_params: None or one entry per image.  Returns the list of 4-panel images."""
        st1 = self._batch_invert_stage(self.ldm_stable, image_paths, prompts_src, prompts_tar)
        return self._batch_edit_stage(st1, prompts_src, prompts_tar, guidance_scale, cross_replace_steps, self_replace_steps, blend_words,
                                      eq_params, is_replace_controller, add_target, return_stages)

    def _batch_invert_stage(self, model, image_paths, prompts_src, prompts_tar):
        """Stage 1 of a batch on `model` (the main pipeline, or the second context of `edit_stream_images_directinversion`): load, prompt
        embedding, VAE encode (+ the reference's discarded decode, inversion.py:357), the 50 n-row inversion steps."""
        eng = model.engine
        n = len(image_paths)
        if not (len(prompts_src) == len(prompts_tar) == n):
            raise ValueError("image_paths, prompts_src and prompts_tar must have the same length")
        if self.ldm_stable.engine.max_unet_rows < 12 * n or eng.max_unet_rows < n:
            raise ValueError(f"batch of {n} images needs a pipeline built with max_unet_rows >= {12 * n}")
        side = eng.cfg.sample_size * eng.cfg.vae_scale
        images_gt = []
        for pth in image_paths:
            im = load_512(pth)
            if side != 512:
                im = np.array(Image.fromarray(im).resize((side, side)))
            images_gt.append(im)
        model.scheduler.set_timesteps(self.num_ddim_steps)
        ts = model.scheduler.timesteps.numpy()
        inv = DirectInversion(model=model, num_ddim_steps=self.num_ddim_steps)
        contexts = []
        register_attention_control(model, None)
        for i in range(n):
            inv.init_prompt([prompts_src[i], prompts_tar[i]])
            contexts.append(inv.context)
        ctx = torch.stack(contexts)                                        # [n, 4, 77, 768]
        z0 = image2latent(model.vae, np.stack(images_gt))                  # [n, 4, h, w]
        latent2image(model.vae, z0)                                        # the reference's discarded image_rec decode (inversion.py:357)
        x_stars = eng.ddim_invert(z0, ctx[:, 2], ts)                       # [steps+1, n, 4, h, w]
        return images_gt, side, ctx, x_stars

    def _batch_edit_stage(self, st1, prompts_src, prompts_tar, guidance_scale, cross_replace_steps, self_replace_steps, blend_words,
                          eq_params, is_replace_controller, add_target, return_stages):
        """Stage 2 on the main pipeline: controllers, the 12n-row lock-step loop from the inversion trajectories, the decodes, the panels."""
        images_gt, side, ctx, x_stars = st1
        model, eng = self.ldm_stable, self.ldm_stable.engine
        n = len(images_gt)
        model.scheduler.set_timesteps(self.num_ddim_steps)
        ts = model.scheduler.timesteps.numpy()
        register_attention_control(model, None)
        controllers = []
        for i in range(n):
            controllers.append(make_controller(pipeline=model, prompts=[prompts_src[i], prompts_tar[i]], is_replace_controller=is_replace_controller,
                                               cross_replace_steps={"default_": cross_replace_steps},
                                               self_replace_steps=self_replace_steps,
                                               blend_words=blend_words[i] if blend_words is not None else None,
                                               equilizer_params=eq_params[i] if eq_params is not None else None,
                                               num_ddim_steps=self.num_ddim_steps, device=self.device))
        nl, lats = eng.direct_edit(x_stars, ctx, [None, [c.tables() for c in controllers]], ts, guidance_scale,
                                   offset_rows=2 if add_target else 1)
        for c in controllers:
            c.cur_step += self.num_ddim_steps
        rec_images = latent2image(model.vae, lats[0].reshape(2 * n, *lats.shape[3:]))
        images = latent2image(model.vae, lats[1].reshape(2 * n, *lats.shape[3:]))
        panels = []
        for i in range(n):
            instruct = txt_draw(f"source prompt: {prompts_src[i]}\ntarget prompt: {prompts_tar[i]}", target_size=(side, side))
            panels.append(Image.fromarray(np.concatenate((instruct, images_gt[i], rec_images[2 * i], images[2 * i + 1]), axis=1)))
        if return_stages:
            return panels, dict(x_stars=x_stars, noise_loss=nl, reconstruct_latents=lats[0], latents=lats[1])
        return panels

    def edit_stream_images_directinversion(self, batches, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
                                           is_replace_controller=False):
        """`edit_images_directinversion` over a sequence of BATCHES with stage overlap (BASELINE config 3's launch shape: batch = 8 per GPU):
        while batch i runs its 50 lock-step steps of 12n rows on the main context, batch i + 1's prompt embeddings, VAE encodes and 50
        n-row inversion steps run on the second context / HIP stream from a worker thread -- what `edit_stream_directinversion` does for
        single images.  Same kernels on the same inputs as `edit_images_directinversion` batch by batch -> identical panels
        (tests/test_gpu_sd1_extras.py); only sweep throughput changes (run_editing_p2p.py:102-146 visits images one by one).
        batches: iterable of (image_paths, prompts_src, prompts_tar, blend_words | None, eq_params | None).  Generator of panel lists."""
        from concurrent.futures import ThreadPoolExecutor
        batches = list(batches)
        if not batches:
            return
        main = self.ldm_stable
        inverter = self._second_pipeline(rows=max(len(b[0]) for b in batches))

        def stage1(b):
            with torch.no_grad(), torch.cuda.device(main.device), torch.cuda.stream(self._inv_stream):
                images_gt, side, ctx, x = self._batch_invert_stage(inverter, b[0], b[1], b[2])
                ctx, x = ctx.clone(), x.clone()
                self._inv_stream.synchronize()          # the consumer is another stream: hand over finished tensors
            return images_gt, side, ctx, x

        with ThreadPoolExecutor(max_workers=1) as ex, torch.no_grad():
            fut = ex.submit(stage1, batches[0])
            for i, b in enumerate(batches):
                st1 = fut.result()
                for t in st1[2:]:
                    if t.is_cuda:                        # produced on the worker's stream, consumed on this one
                        t.record_stream(torch.cuda.current_stream(main.device))
                if i + 1 < len(batches):
                    fut = ex.submit(stage1, batches[i + 1])
                yield self._batch_edit_stage(st1, b[1], b[2], guidance_scale, cross_replace_steps, self_replace_steps,
                                             b[3] if len(b) > 3 else None, b[4] if len(b) > 4 else None, is_replace_controller, False, False)

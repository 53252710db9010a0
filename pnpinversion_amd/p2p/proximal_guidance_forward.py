"""proximal_guidance_forward with the signature of models/p2p/proximal_guidance_forward.py:84-170.

What the reference's editors can reach (p2p_editor.py:324-413, 550-638; run_editing_p2p.py:286-300 passes proximal="l0",
quantile=0.75, use_inversion_guidance=True, recon_lr=1, recon_t=400):
  * the proximal step itself (:39-64): the CFG difference is soft-thresholded at a quantile of its magnitude (device quantile +
    shrink fused into the CFG / DDIM-step kernel);
  * reconstruction guidance (image_enc given, :48-51,60-72 + DDIMSchedulerDev.step's ref_image branch, scheduler_dev.py:68-76; the
    editors pass it with use_reconstruction_guidance=True): at t inside the recon_t window the predicted x0 is pulled towards the
    encoded source image outside the dilated edit mask -- same kernel (pnpi_recon_desc);
  * inversion guidance (:73-75): `mask_edit is not None and inversion_guidance and (recon_t > 0 and t < recon_t) or (recon_t < 0 and
    t > -recon_t)` -- by operator precedence the pull of the step's result towards x_stars[len(x_stars) - i - 2] outside the edit mask
    runs (a) with inversion_guidance=True inside a positive recon_t window, and (b) ALWAYS inside a negative recon_t window (t > -recon_t),
    flag or not -- where the reference then needs x_stars and a mask (it raises TypeError on `1 - None` otherwise; here: ValueError for
    missing x_stars, TypeError for a negative recon_t without edit_stage / prox, both before the loop starts).
    Same kernel (pnpi_recon_desc::inv_x_stars).  No reference editor passes inversion_guidance=True or a negative recon_t
    (p2p_editor.py:368,593 hard-code False); tests/golden/proximal_inv_guidance.npz is a direct call of the reference's function."""
import torch

from .p2p_guidance_forward import p2p_guidance_forward


@torch.no_grad()
def proximal_guidance_forward(model, prompt, controller, guidance_scale=7.5, generator=None, latent=None, uncond_embeddings=None,
                              edit_stage=True, prox=None, quantile=0.7, image_enc=None, recon_lr=0.1, recon_t=400,
                              inversion_guidance=False, x_stars=None, dilate_mask=None, num_inference_steps=None):
    if edit_stage and prox is not None and prox not in ("l0", "l1"):
        raise NotImplementedError
    recon = None
    if edit_stage and prox is not None:
        pull = (inversion_guidance and recon_t > 0) or recon_t < 0          # the reference's precedence (see above)
        if pull and x_stars is None:
            raise ValueError("inversion guidance (inversion_guidance=True, or any negative recon_t) needs x_stars")
        # the pred-x0 pull runs for recon_lr > 0 only (scheduler_dev.py:68); the inversion pull is `latents - recon_lr * (...)` with no such
        # test (:75): any non-zero recon_lr, negative ones included
        if (image_enc is not None and recon_lr > 0) or (pull and recon_lr != 0):
            recon = dict(ref_image=image_enc if recon_lr > 0 else None, recon_lr=recon_lr, recon_t=recon_t, dilate_mask=dilate_mask or 0,
                         x_stars=x_stars if pull else None)
    steps = num_inference_steps if num_inference_steps is not None else model.scheduler.num_inference_steps
    if recon_t < 0 and not (edit_stage and prox is not None):
        # the reference's line 73 is true at every step with t > -recon_t whatever the other flags say, and then evaluates `1 - mask_edit`
        # with mask_edit = None (no proximal step ran): TypeError at the first such step.  Same outcome here, stated up front.
        max_t = (steps - 1) * (model.scheduler.config.num_train_timesteps // steps) + model.scheduler.config.steps_offset
        if max_t > -recon_t:
            raise TypeError("negative recon_t without edit_stage / prox: the reference evaluates `1 - mask_edit` with mask_edit = None "
                            "(models/p2p/proximal_guidance_forward.py:73-74)")
    return p2p_guidance_forward(model=model, prompt=prompt, controller=controller, num_inference_steps=steps,
                                guidance_scale=guidance_scale, generator=generator, latent=latent, uncond_embeddings=uncond_embeddings,
                                prox=prox if edit_stage else None, quantile=quantile, recon=recon)

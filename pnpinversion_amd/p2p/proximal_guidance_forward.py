"""proximal_guidance_forward with the signature of models/p2p/proximal_guidance_forward.py:84-170.

What the reference's editors can reach (p2p_editor.py:324-413, 550-638; run_editing_p2p.py:286-300 passes proximal="l0",
quantile=0.75, use_inversion_guidance=True, recon_lr=1, recon_t=400):
  * the proximal step itself (:39-64): the CFG difference is soft-thresholded at a quantile of its magnitude (device quantile +
    shrink fused into the CFG / DDIM-step kernel);
  * reconstruction guidance (image_enc given, :48-51,60-72 + DDIMSchedulerDev.step's ref_image branch, scheduler_dev.py:68-76; the
    editors pass it with use_reconstruction_guidance=True): at t inside the recon_t window the predicted x0 is pulled towards the
    encoded source image outside the dilated edit mask -- same kernel (pnpi_recon_desc);
  * inversion guidance (:73-75): the editors never pass inversion_guidance=True to this function, and with the default False the
    condition `mask_edit is not None and inversion_guidance and (...) or (recon_t < 0 and ...)` is False for recon_t > 0 -- inert."""
import torch

from .p2p_guidance_forward import p2p_guidance_forward


@torch.no_grad()
def proximal_guidance_forward(model, prompt, controller, guidance_scale=7.5, generator=None, latent=None, uncond_embeddings=None,
                              edit_stage=True, prox=None, quantile=0.7, image_enc=None, recon_lr=0.1, recon_t=400,
                              inversion_guidance=False, x_stars=None, dilate_mask=None, num_inference_steps=None):
    if edit_stage and prox is not None and prox not in ("l0", "l1"):
        raise NotImplementedError
    if inversion_guidance or recon_t < 0:
        raise NotImplementedError("inversion guidance / negative recon_t are not built (no reference editor passes them)")
    recon = None
    if edit_stage and prox is not None and image_enc is not None and recon_lr > 0:
        recon = dict(ref_image=image_enc, recon_lr=recon_lr, recon_t=recon_t, dilate_mask=dilate_mask or 0)
    steps = num_inference_steps if num_inference_steps is not None else model.scheduler.num_inference_steps
    return p2p_guidance_forward(model=model, prompt=prompt, controller=controller, num_inference_steps=steps,
                                guidance_scale=guidance_scale, generator=generator, latent=latent, uncond_embeddings=uncond_embeddings,
                                prox=prox if edit_stage else None, quantile=quantile, recon=recon)

"""proximal_guidance_forward with the signature of models/p2p/proximal_guidance_forward.py:84-170.  With prox=None -- how
`negative-prompt-inversion+p2p` calls it (p2p_editor.py:59-66) -- every proximal / reconstruction-guidance branch is inert
(mask_edit stays None, the scheduler gets recon_lr = 0) and the loop is the plain P2P CFG loop.  The l0 / l1 proximal variants
need a device quantile + dilate and are not built."""
import torch

from .p2p_guidance_forward import p2p_guidance_forward


@torch.no_grad()
def proximal_guidance_forward(model, prompt, controller, guidance_scale=7.5, generator=None, latent=None, uncond_embeddings=None,
                              edit_stage=True, prox=None, quantile=0.7, image_enc=None, recon_lr=0.1, recon_t=400,
                              inversion_guidance=False, x_stars=None, dilate_mask=None, num_inference_steps=None):
    if edit_stage and prox is not None:
        raise NotImplementedError("proximal guidance (prox = %r) is not built (SURVEY 8f rank 3)" % (prox,))
    steps = num_inference_steps if num_inference_steps is not None else model.scheduler.num_inference_steps
    return p2p_guidance_forward(model=model, prompt=prompt, controller=controller, num_inference_steps=steps,
                                guidance_scale=guidance_scale, generator=generator, latent=latent, uncond_embeddings=uncond_embeddings)

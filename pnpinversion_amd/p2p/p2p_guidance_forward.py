"""direct_inversion_p2p_guidance_forward with the signature of models/p2p/p2p_guidance_forward.py:135-173.
The 50-step dual-branch CFG loop (UNet B=4 -> CFG -> DDIM step -> source-branch offset -> controller.step_callback /
LocalBlend) runs as one device-resident pnpi_edit_loop call."""
import torch

from ..utils.utils import init_latent
from .attention_control import controller_tables, is_callback_controller, register_attention_control


def _encode_prompts(model, prompt):
    tok = model.tokenizer
    text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    text_embeddings = model.text_encoder(text_input.input_ids.to(model.device))[0]
    max_length = text_input.input_ids.shape[-1]
    uncond_input = tok([""] * len(prompt), padding="max_length", max_length=max_length, return_tensors="pt")
    uncond_embeddings = model.text_encoder(uncond_input.input_ids.to(model.device))[0]
    return torch.cat([uncond_embeddings, text_embeddings])            # [unc_src, unc_tgt, cond_src, cond_tgt]


def _level1_loop(model, controller, latents, context_of_step, guidance_scale, noise_loss_list, offset_rows):
    """The reference's Python step loop (p2p_guidance_forward.py:21-62,103-116,135-173) for a controller of the call-back protocol
    (no kernel descriptor): per step one 4-row model.unet(...) call -- NativeUNet materialises the probabilities of the 32 attention
    sites and calls `controller(attn, is_cross, place)` at each -- then the fused CFG / DDIM-step / offset kernel and the controller's
    own step_callback.  Slow and exact; the device-resident loops are for controllers with a descriptor."""
    ratio = model.scheduler.step_ratio
    for i, t in enumerate(model.scheduler.timesteps):
        eps = model.unet(torch.cat([latents] * 2), t, encoder_hidden_states=context_of_step(i))["sample"]
        nl = noise_loss_list[i] if noise_loss_list is not None else None
        latents = model.engine.cfg_ddim_prev(eps, latents, int(t), ratio, guidance_scale, noise_loss=nl, offset_rows=offset_rows)
        latents = controller.step_callback(latents)
    return latents


def _run(model, prompt, controller, latent, num_inference_steps, guidance_scale, generator, noise_loss_list, add_offset,
         offset_rows):
    batch_size = len(prompt)
    if batch_size != 2:
        raise NotImplementedError("the native loop handles one (source, target) prompt pair per image")
    register_attention_control(model, controller)
    height = width = model.engine.cfg.sample_size * model.engine.cfg.vae_scale
    context = _encode_prompts(model, prompt)
    latent, latents = init_latent(latent, model, height, width, generator, batch_size)
    model.scheduler.set_timesteps(num_inference_steps)
    ctrl = model.unet.controller          # the registered form (a foreign controller object may have been adapted)
    if is_callback_controller(ctrl):
        out = _level1_loop(model, ctrl, latents, lambda i: context, guidance_scale, noise_loss_list if add_offset else None, offset_rows)
        return out, latent
    tables = controller_tables(ctrl)
    nl = None
    if noise_loss_list is not None and add_offset:
        nl = torch.stack(list(noise_loss_list))[:, None]               # [steps, 1, 2, 4, h, w]
    out = model.engine.edit_loop(latent.reshape(1, *latent.shape[-3:]), context[None], nl, [tables] if tables is not None else None,
                                 model.scheduler.timesteps.numpy(), guidance_scale, offset_rows=offset_rows)
    if controller is not None and hasattr(controller, "cur_step"):
        controller.cur_step += num_inference_steps
    return out[0], latent


@torch.no_grad()
def direct_inversion_p2p_guidance_forward(model, prompt, controller, latent=None, num_inference_steps: int = 50,
                                          guidance_scale=7.5, generator=None, noise_loss_list=None, add_offset=True):
    return _run(model, prompt, controller, latent, num_inference_steps, guidance_scale, generator, noise_loss_list, add_offset, 1)


@torch.no_grad()
def direct_inversion_p2p_guidance_forward_add_target(model, prompt, controller, latent=None, num_inference_steps: int = 50,
                                                     guidance_scale=7.5, generator=None, noise_loss_list=None, add_offset=True):
    """p2p_guidance_forward.py:119-132,175-213: the offset is added to both branches."""
    return _run(model, prompt, controller, latent, num_inference_steps, guidance_scale, generator, noise_loss_list, add_offset, 2)


def _constant_uncond(uncond_embeddings):
    """ddim+p2p and negative-prompt inversion hand over the same embedding for every step (inversion.py:226, 98): one context for
    the whole loop (text K / V projected once).  Returns None for a genuinely per-step list (null-text optimisation)."""
    first = uncond_embeddings[0]
    for u in uncond_embeddings[1:]:
        if u is not first and not torch.equal(u, first):
            return None
    return first


@torch.no_grad()
def p2p_guidance_forward(model, prompt, controller, num_inference_steps: int = 50, guidance_scale=7.5, generator=None, latent=None,
                         uncond_embeddings=None, prox=None, quantile=0.7, recon=None, single_branch=False):
    """models/p2p/p2p_guidance_forward.py:21-62: the plain Prompt-to-Prompt CFG loop (no direct-inversion offset); one or two
    prompts; `uncond_embeddings` = per-step replacement of the "" embedding."""
    batch_size = len(prompt)
    if batch_size not in (1, 2):
        raise NotImplementedError("the native loop handles one prompt or one (source, target) pair")
    register_attention_control(model, controller)
    height = width = model.engine.cfg.sample_size * model.engine.cfg.vae_scale
    tok = model.tokenizer
    text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    text = model.text_encoder(text_input.input_ids.to(model.device))[0]
    per_step = None
    const = _constant_uncond(uncond_embeddings) if uncond_embeddings is not None else None
    if uncond_embeddings is None or const is None or single_branch:
        uncond_input = tok([""] * batch_size, padding="max_length", max_length=text_input.input_ids.shape[-1], return_tensors="pt")
        uncond = model.text_encoder(uncond_input.input_ids.to(model.device))[0]
        if uncond_embeddings is not None:
            per_step = torch.stack([u.to(text.device).reshape(1, *u.shape[-2:]) for u in uncond_embeddings])     # [steps, 1, 77, D]
    else:
        uncond = const.to(text.device).expand(*text.shape)
    latent, latents = init_latent(latent, model, height, width, generator, batch_size)
    model.scheduler.set_timesteps(num_inference_steps)
    if batch_size == 1:   # the kernel batch is [unc_a, unc_b, cond_a, cond_b]: run the single prompt as both rows of a pair
        uncond, text = uncond.expand(2, *uncond.shape[1:]), text.expand(2, *text.shape[1:])
    context = torch.cat([uncond, text])
    ctrl = model.unet.controller          # the registered form (a foreign controller object may have been adapted)
    if is_callback_controller(ctrl):
        if prox is not None or recon is not None or batch_size != 2:
            raise NotImplementedError("call-back controllers run the plain two-prompt guidance loop only (no proximal step, no reconstruction guidance)")
        def context_of_step(i):
            if per_step is None:
                return context
            c = context.clone()
            c[:1 if single_branch else 2] = per_step[i].to(context.device)
            return c
        return _level1_loop(model, ctrl, latents, context_of_step, guidance_scale, None, 0), latent
    tables = controller_tables(ctrl)
    if prox is not None and batch_size != 2:
        raise NotImplementedError("the proximal step takes its quantile over the (source, target) pair")
    if per_step is not None:      # null-text inversion: the step's embedding on every unconditional row (:56-57) or the first only (:92)
        out = model.engine.edit_loop_uncond_steps(latent.reshape(1, *latent.shape[-3:]), context[None], per_step,
                                                  [tables] if tables is not None else None, model.scheduler.timesteps.numpy(), guidance_scale,
                                                  first_only=single_branch, prox=prox, quantile=quantile, recon=recon)
    else:
        out = model.engine.edit_loop(latent.reshape(1, *latent.shape[-3:]), context[None], None, [tables] if tables is not None else None,
                                     model.scheduler.timesteps.numpy(), guidance_scale, prox=prox, quantile=quantile, recon=recon)
    if controller is not None and hasattr(controller, "cur_step"):
        controller.cur_step += num_inference_steps
    return out[0][:batch_size], latent



def p2p_guidance_forward_single_branch(model, prompt, controller, num_inference_steps: int = 50, guidance_scale=7.5, generator=None,
                                       latent=None, uncond_embeddings=None):
    """models/p2p/p2p_guidance_forward.py:65-100: the optimised embedding replaces the "" embedding of the FIRST prompt only."""
    return p2p_guidance_forward(model, prompt, controller, num_inference_steps, guidance_scale, generator, latent, uncond_embeddings,
                                single_branch=True)

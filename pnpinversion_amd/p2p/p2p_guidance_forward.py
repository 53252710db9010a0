"""direct_inversion_p2p_guidance_forward with the signature of models/p2p/p2p_guidance_forward.py:135-173.
The 50-step dual-branch CFG loop (UNet B=4 -> CFG -> DDIM step -> source-branch offset -> controller.step_callback /
LocalBlend) runs as one device-resident pnpi_edit_loop call."""
import torch

from ..utils.utils import init_latent
from .attention_control import register_attention_control


def _encode_prompts(model, prompt):
    tok = model.tokenizer
    text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    text_embeddings = model.text_encoder(text_input.input_ids.to(model.device))[0]
    max_length = text_input.input_ids.shape[-1]
    uncond_input = tok([""] * len(prompt), padding="max_length", max_length=max_length, return_tensors="pt")
    uncond_embeddings = model.text_encoder(uncond_input.input_ids.to(model.device))[0]
    return torch.cat([uncond_embeddings, text_embeddings])            # [unc_src, unc_tgt, cond_src, cond_tgt]


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
    tables = controller.tables() if controller is not None and hasattr(controller, "tables") else None
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

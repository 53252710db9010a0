"""DirectInversion with the interface of models/p2p/inversion.py:245-400, driving the device-resident loops of libpnpi.

    invert(image_gt, prompt, guidance_scale) -> (image_gt, image_rec, ddim_latents, noise_loss_list)

Faithful schedule: 50 B=1 UNet calls (ddim_loop, :308-319) + 50 B=4 calls (offset_calculate, :375-391), one VAE encode and the
VAE decode whose result the reference's caller throws away (:357).  Each phase is ONE C-ABI call; no host round trip per step."""
import torch

from ..utils.utils import image2latent, latent2image, slerp_tensor
from .attention_control import register_attention_control


class DirectInversion:
    def __init__(self, model, num_ddim_steps):
        self.model = model
        self.tokenizer = model.tokenizer
        self.prompt = None
        self.context = None
        self.num_ddim_steps = num_ddim_steps

    @property
    def scheduler(self):
        return self.model.scheduler

    @property
    def _engine(self):
        return self.model.engine

    # ---- single steps (inversion.py:247-270), exposed for API parity
    def prev_step(self, model_output, timestep, sample):
        return self._engine.ddim_prev_step(model_output, int(timestep), self.scheduler.step_ratio, sample), None

    def next_step(self, model_output, timestep, sample):
        return self._engine.ddim_next_step(model_output, int(timestep), self.scheduler.step_ratio, sample)

    def get_noise_pred_single(self, latents, t, context):
        return self.model.unet(latents, t, encoder_hidden_states=context)["sample"]

    @torch.no_grad()
    def init_prompt(self, prompt):
        """inversion.py:290-306: context = cat([uncond x len(prompt), text])"""
        tok = self.model.tokenizer
        uncond_input = tok([""] * len(prompt), padding="max_length", max_length=tok.model_max_length, return_tensors="pt")
        uncond = self.model.text_encoder(uncond_input.input_ids.to(self.model.device))[0]
        text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
        text = self.model.text_encoder(text_input.input_ids.to(self.model.device))[0]
        self.context = torch.cat([uncond, text])
        self.prompt = prompt

    @torch.no_grad()
    def ddim_loop(self, latent):
        """inversion.py:308-319 -> list of num_ddim_steps + 1 latents [1,4,h,w]"""
        uncond, cond = self.context.chunk(2)
        all_lat = self._engine.ddim_invert(latent, cond[[0]], self.scheduler.timesteps.numpy())
        return [all_lat[i] for i in range(all_lat.shape[0])]

    @torch.no_grad()
    def ddim_inversion(self, image):
        latent = image2latent(self.model.vae, image)
        image_rec = latent2image(self.model.vae, latent)[0]
        ddim_latents = self.ddim_loop(latent)
        return image_rec, ddim_latents

    def offset_calculate(self, latents, num_inner_steps, epsilon, guidance_scale):
        """inversion.py:375-391 -> list of num_ddim_steps tensors [2,4,h,w]"""
        if self.context.shape[0] != 4:
            raise NotImplementedError("offset_calculate handles one (source, target) prompt pair")
        lat = torch.stack(latents)                               # [steps+1, 1, 4, h, w]
        nl = self._engine.offset_calculate(lat, self.context[None], self.scheduler.timesteps.numpy(), guidance_scale)
        return [nl[i, 0] for i in range(nl.shape[0])]

    def invert(self, image_gt, prompt, guidance_scale, num_inner_steps=10, early_stop_epsilon=1e-5):
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        noise_loss_list = self.offset_calculate(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale)
        return image_gt, image_rec, ddim_latents, noise_loss_list

    # ---- ablation variants (inversion.py:334-347, 366-373, 407-412, 478-535): same loops, one knob each
    @torch.no_grad()
    def ddim_with_guidance_scale_loop(self, latent, guidance_scale):
        """inversion.py:334-347: inversion under CFG (uncond and cond rows in one launch per step)"""
        uncond, cond = self.context.chunk(2)
        all_lat = self._engine.ddim_invert_cfg(latent, uncond[[0]], cond[[0]], self.scheduler.timesteps.numpy(), guidance_scale)
        return [all_lat[i] for i in range(all_lat.shape[0])]

    @torch.no_grad()
    def ddim_with_guidance_scale_inversion(self, image, guidance_scale):
        latent = image2latent(self.model.vae, image)
        image_rec = latent2image(self.model.vae, latent)[0]
        return image_rec, self.ddim_with_guidance_scale_loop(latent, guidance_scale)

    def _offsets(self, latents, guidance_scale, offset_scale):
        if self.context.shape[0] != 4:
            raise NotImplementedError("offset_calculate handles one (source, target) prompt pair")
        nl = self._engine.offset_calculate(torch.stack(latents), self.context[None], self.scheduler.timesteps.numpy(), guidance_scale,
                                           offset_scale=offset_scale)
        return [nl[i, 0] for i in range(nl.shape[0])]

    def offset_calculate_not_full(self, latents, num_inner_steps, epsilon, guidance_scale, scale):
        """inversion.py:478-493: loss = (x*_{t-1} - prev_rec) * scale"""
        return self._offsets(latents, guidance_scale, float(scale))

    def offset_calculate_skip_step(self, latents, num_inner_steps, epsilon, guidance_scale, skip_step):
        """inversion.py:502-519: the offset is applied on steps i % skip_step == 0 and is zero otherwise"""
        return self._offsets(latents, guidance_scale, [1.0 if i % skip_step == 0 else 0.0 for i in range(self.num_ddim_steps)])

    def invert_with_guidance_scale_vary_guidance(self, image_gt, prompt, inverse_guidance_scale, forward_guidance_scale,
                                                 num_inner_steps=10, early_stop_epsilon=1e-5):
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents = self.ddim_with_guidance_scale_inversion(image_gt, inverse_guidance_scale)
        noise_loss_list = self.offset_calculate(ddim_latents, num_inner_steps, early_stop_epsilon, forward_guidance_scale)
        return image_gt, image_rec, ddim_latents, noise_loss_list

    def invert_not_full(self, image_gt, prompt, guidance_scale, num_inner_steps=10, early_stop_epsilon=1e-5, scale=1.):
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        noise_loss_list = self.offset_calculate_not_full(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale, scale)
        return image_gt, image_rec, ddim_latents, noise_loss_list

    def invert_skip_step(self, image_gt, prompt, guidance_scale, skip_step, num_inner_steps=10, early_stop_epsilon=1e-5, scale=1.):
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        noise_loss_list = self.offset_calculate_skip_step(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale, skip_step)
        return image_gt, image_rec, ddim_latents, noise_loss_list

    def null_latent_calculate(self, latents, num_inner_steps, epsilon, guidance_scale):
        """inversion.py:419-460 -> list of num_ddim_steps tensors [2,4,h,w]: the device loop pnpi_null_latent_calculate (per step the
        null-text optimisation of the source row's unconditional embedding, then the step's effect as a latent offset)."""
        if self.context.shape[0] != 4:
            raise NotImplementedError("null_latent_calculate handles one (source, target) prompt pair")
        nl, self.inner_iterations, self.inner_losses = self._engine.null_latent_calculate(
            torch.stack(list(latents)), self.context, self.scheduler.timesteps.numpy(), guidance_scale, num_inner_steps, epsilon,
            return_losses=True)
        return [nl[i] for i in range(nl.shape[0])]

    def invert_null_latent(self, image_gt, prompt, guidance_scale, num_inner_steps=10, early_stop_epsilon=1e-5):
        """inversion.py:462-470"""
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        latent_list = self.null_latent_calculate(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale)
        return image_gt, image_rec, ddim_latents, latent_list

    def invert_without_attn_controller(self, image_gt, prompt, guidance_scale, num_inner_steps=10, early_stop_epsilon=1e-5):
        self.init_prompt(prompt)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        noise_loss_list = self.offset_calculate(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale)
        return image_gt, image_rec, ddim_latents, noise_loss_list


class _SinglePromptInversion:
    """Shared part of NullInversion / NegativePromptInversion (inversion.py:10-108, 110-242): one prompt, context = [uncond, cond]."""

    def __init__(self, model, num_ddim_steps):
        self.model = model
        self.tokenizer = model.tokenizer
        self.prompt = None
        self.context = None
        self.num_ddim_steps = num_ddim_steps

    @property
    def scheduler(self):
        return self.model.scheduler

    def prev_step(self, model_output, timestep, sample):
        return self.model.engine.ddim_prev_step(model_output, int(timestep), self.scheduler.step_ratio, sample)

    def next_step(self, model_output, timestep, sample):
        return self.model.engine.ddim_next_step(model_output, int(timestep), self.scheduler.step_ratio, sample)

    def get_noise_pred_single(self, latents, t, context):
        return self.model.unet(latents, t, encoder_hidden_states=context)["sample"]

    @torch.no_grad()
    def init_prompt(self, prompt: str):
        tok = self.model.tokenizer
        uncond_input = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt")
        uncond = self.model.text_encoder(uncond_input.input_ids.to(self.model.device))[0]
        text_input = tok([prompt], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
        text = self.model.text_encoder(text_input.input_ids.to(self.model.device))[0]
        self.context = torch.cat([uncond, text])
        self.prompt = prompt

    @torch.no_grad()
    def ddim_loop(self, latent):
        uncond, cond = self.context.chunk(2)
        all_lat = self.model.engine.ddim_invert(latent, cond, self.scheduler.timesteps.numpy())
        return [all_lat[i] for i in range(all_lat.shape[0])]


class NullInversion(_SinglePromptInversion):
    """inversion.py:110-242.  num_inner_steps = 0 is what `ddim+p2p` uses (p2p_editor.py:155-156); > 0 is null-text inversion."""

    @torch.no_grad()
    def ddim_inversion(self, image):
        latent = image2latent(self.model.vae, image)
        image_rec = latent2image(self.model.vae, latent)[0]
        return image_rec, self.ddim_loop(latent)

    def null_optimization(self, latents, num_inner_steps, epsilon, guidance_scale):
        """inversion.py:196-225: the device loop pnpi_null_text_optimize (recording UNet forward, reverse walk to the embedding, Adam)."""
        uncond, cond = self.context.chunk(2)
        if num_inner_steps == 0:
            # the embedding never changes; the reference also walks latent_cur down a 50-step CFG loop whose result nobody reads
            # (inversion.py:228-230) -- not executed here
            return [uncond[:1]] * self.num_ddim_steps
        embs, self.inner_iterations, self.inner_losses = self.model.engine.null_text_optimize(
            torch.stack(list(latents)), uncond, cond, self.scheduler.timesteps.numpy(), guidance_scale, num_inner_steps, epsilon,
            return_losses=True)
        return [embs[i] for i in range(embs.shape[0])]

    def invert(self, image_gt, prompt, guidance_scale, num_inner_steps=10, early_stop_epsilon=1e-5):
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        uncond_embeddings = self.null_optimization(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale)
        return image_gt, image_rec, ddim_latents, uncond_embeddings


class NegativePromptInversion(_SinglePromptInversion):
    """inversion.py:10-108: the 'unconditional' embedding of every step is the source prompt's conditional embedding."""

    @torch.no_grad()
    def ddim_inversion(self, image):
        latent = image2latent(self.model.vae, image)
        image_rec = latent2image(self.model.vae, latent)[0]
        return image_rec, self.ddim_loop(latent), latent

    def invert(self, image_gt, prompt, npi_interp=0.0):
        self.init_prompt(prompt)
        register_attention_control(self.model, None)
        image_rec, ddim_latents, image_rec_latent = self.ddim_inversion(image_gt)
        uncond, cond = self.context.chunk(2)
        if npi_interp > 0.0:     # inversion.py:98-99: vector interpolation between the conditional and the unconditional embedding
            cond = slerp_tensor(npi_interp, cond, uncond)
        return image_rec, image_rec_latent, ddim_latents, [cond] * self.num_ddim_steps

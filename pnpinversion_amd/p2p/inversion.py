"""DirectInversion with the interface of models/p2p/inversion.py:245-400, driving the device-resident loops of libpnpi.

    invert(image_gt, prompt, guidance_scale) -> (image_gt, image_rec, ddim_latents, noise_loss_list)

Faithful schedule: 50 B=1 UNet calls (ddim_loop, :308-319) + 50 B=4 calls (offset_calculate, :375-391), one VAE encode and the
VAE decode whose result the reference's caller throws away (:357).  Each phase is ONE C-ABI call; no host round trip per step."""
import torch

from ..utils.utils import image2latent, latent2image
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

    def invert_without_attn_controller(self, image_gt, prompt, guidance_scale, num_inner_steps=10, early_stop_epsilon=1e-5):
        self.init_prompt(prompt)
        image_rec, ddim_latents = self.ddim_inversion(image_gt)
        noise_loss_list = self.offset_calculate(ddim_latents, num_inner_steps, early_stop_epsilon, guidance_scale)
        return image_gt, image_rec, ddim_latents, noise_loss_list

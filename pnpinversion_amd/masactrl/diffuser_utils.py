"""MasaCtrlPipeline with the call surface of models/masactrl/diffuser_utils.py:9-270 that run_editing_masactrl.py uses:
`__call__(prompt, latents=..., guidance_scale=..., noise_loss_list=...)` -> float images in [0,1] [B,3,H,W], and
`invert(image, prompt, guidance_scale=..., num_inference_steps=..., return_intermediates=True)`.  Both are one device-resident
loop call of libpnpi (pnpi_edit_loop / pnpi_ddim_invert_cfg); the registered attention editor travels as a descriptor."""
import numpy as np
import torch

from ..p2p.attention_control import controller_tables
from ..pipeline import NativePipeline


class MasaCtrlPipeline(NativePipeline):
    masactrl_editor = None

    def _embed(self, prompts):
        tok = self.tokenizer
        ids = tok(prompts, padding="max_length", max_length=77, return_tensors="pt").input_ids
        return self.text_encoder(ids.to(self.device))[0]

    @torch.no_grad()
    def image2latent(self, image):
        """diffuser_utils.py:62-72: image float [-1,1] [1,3,H,W]"""
        return self.vae.encode(image)["latent_dist"].mean * 0.18215

    @torch.no_grad()
    def latent2image(self, latents, return_type="np"):
        """diffuser_utils.py:74-84"""
        image = self.vae.decode(1 / 0.18215 * latents.detach())["sample"]
        image = (image / 2 + 0.5).clamp(0, 1)
        if return_type == "np":
            image = (image.cpu().permute(0, 2, 3, 1).numpy()[0] * 255).astype(np.uint8)
        return image

    @torch.no_grad()
    def __call__(self, prompt, batch_size=1, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, eta=0.0,
                 latents=None, unconditioning=None, neg_prompt=None, ref_intermediate_latents=None, return_intermediates=False,
                 noise_loss_list=None, **kwds):
        """diffuser_utils.py:91-193 for what the editors pass: CFG sampling of one or two prompts from given latents."""
        if isinstance(prompt, str):
            prompt = [prompt] * batch_size
        if latents is None or unconditioning is not None or ref_intermediate_latents is not None or return_intermediates or \
                kwds.get("dir") or not guidance_scale > 1. or len(prompt) not in (1, 2):
            raise NotImplementedError("MasaCtrlPipeline: only the call patterns of run_editing_masactrl.py are built")
        n = len(prompt)
        text = self._embed(prompt)
        uncond = self._embed([neg_prompt if neg_prompt else ""] * n)
        assert latents.shape[0] == n
        if n == 1:   # the kernel batch is [unc_a, unc_b, cond_a, cond_b]: a single prompt runs as both rows of a pair
            text, uncond, latents = text.expand(2, -1, -1), uncond.expand(2, -1, -1), latents.expand(2, -1, -1, -1)
        if not torch.equal(latents[0], latents[1]):
            raise NotImplementedError("the two rows must start from the same latent (as in run_editing_masactrl.py)")
        context = torch.cat([uncond, text])
        self.scheduler.set_timesteps(num_inference_steps)
        ed = self.masactrl_editor
        tables = controller_tables(ed)
        nl = torch.stack(list(noise_loss_list))[:, None] if noise_loss_list is not None else None
        out = self.engine.edit_loop(latents[:1].reshape(1, *latents.shape[-3:]), context[None], nl,
                                    [tables] if tables is not None else None, self.scheduler.timesteps.numpy(), guidance_scale)
        if ed is not None:
            ed.cur_step += num_inference_steps
        return self.latent2image(out[0][:n], return_type="pt")

    @torch.no_grad()
    def invert(self, image, prompt, num_inference_steps=50, guidance_scale=7.5, eta=0.0, return_intermediates=False, **kwds):
        """diffuser_utils.py:195-270: DDIM inversion under CFG with prompt vs "" -> (x_T, [x_0 .. x_T])"""
        if not isinstance(prompt, str):
            raise NotImplementedError("invert() takes one prompt string")
        text, uncond = self._embed([prompt]), self._embed([""])
        latents = self.image2latent(image)
        self.scheduler.set_timesteps(num_inference_steps)
        ts = self.scheduler.timesteps.numpy()
        if guidance_scale > 1.:
            all_lat = self.engine.ddim_invert_cfg(latents, uncond, text, ts, guidance_scale)
        else:
            all_lat = self.engine.ddim_invert(latents, text, ts)
        lst = [all_lat[i] for i in range(all_lat.shape[0])]
        return (lst[-1], lst) if return_intermediates else lst[-1]

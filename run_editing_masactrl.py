#!/usr/bin/env python
"""MasaCtrl editing driver with the CLI and the MasaCtrlEditor class of the reference's run_editing_masactrl.py (methods
"ddim+masactrl" and "directinversion+masactrl"), on the MI355X-native pipeline: inversion, direct-inversion offsets and the
mutual-self-attention sampling loop are device-resident libpnpi loops; the attention editor is a kernel-side row table.  Under
torch.distributed.run the work list is sharded over the ranks exactly like run_editing_p2p.py."""
import argparse
import json
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from pnpinversion_amd.masactrl.diffuser_utils import MasaCtrlPipeline
from pnpinversion_amd.masactrl.masactrl import MutualSelfAttentionControl
from pnpinversion_amd.masactrl.masactrl_utils import AttentionBase, regiter_attention_editor_diffusers
from pnpinversion_amd.p2p.inversion import DirectInversion
from pnpinversion_amd.utils.utils import load_512, txt_draw
from run_editing_p2p import mask_decode, setup_seed  # noqa: F401  (same helpers as the reference's copy, :21-46)


def load_image(image_path, device):
    """run_editing_masactrl.py:49-54: RGB -> float [-1, 1] [1,3,H,W] -> nearest resize to 512 x 512"""
    arr = np.array(Image.open(image_path).convert("RGB")) if isinstance(image_path, str) else np.asarray(image_path)[:, :, :3]
    image = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)[:3].unsqueeze(0).float() / 127.5 - 1.
    return F.interpolate(image, (512, 512)).to(device)


class MasaCtrlEditor:
    """run_editing_masactrl.py:57-175"""

    def __init__(self, method_list, device, num_ddim_steps=50, *, pipeline=None, weight_seed=0):
        self.device = device
        self.method_list = method_list
        self.num_ddim_steps = num_ddim_steps
        if pipeline is None:   # the reference loads CompVis/stable-diffusion-v1-4 here; no checkpoint exists offline
            pipeline = MasaCtrlPipeline.synthetic(seed=weight_seed, device=device)
        self.model = pipeline
        self.scheduler = pipeline.scheduler
        self.model.scheduler.set_timesteps(self.num_ddim_steps)

    def __call__(self, edit_method, image_path, prompt_src, prompt_tar, guidance_scale, step=4, layper=10):
        if edit_method == "ddim+masactrl":
            return self.edit_image_ddim_MasaCtrl(image_path, prompt_src, prompt_tar, guidance_scale, step=step, layper=layper)
        elif edit_method == "directinversion+masactrl":
            return self.edit_image_directinversion_MasaCtrl(image_path, prompt_src, prompt_tar, guidance_scale, step=step, layper=layper)
        raise NotImplementedError(f"No edit method named {edit_method}")

    def _side(self):
        return self.model.engine.cfg.sample_size * self.model.engine.cfg.vae_scale

    def _sample(self, prompt_tar, start, guidance_scale, step, layper, noise_loss_list):
        prompts = ["", prompt_tar]
        regiter_attention_editor_diffusers(self.model, AttentionBase())
        image_fixed = self.model([prompt_tar], latents=start[-1:], num_inference_steps=self.num_ddim_steps,
                                 guidance_scale=guidance_scale)                                   # "direct synthesis" (:104-110)
        regiter_attention_editor_diffusers(self.model, MutualSelfAttentionControl(step, layper, total_steps=max(50, self.num_ddim_steps)))
        image_masactrl = self.model(prompts, latents=start, num_inference_steps=self.num_ddim_steps, guidance_scale=guidance_scale,
                                    noise_loss_list=noise_loss_list)
        return image_fixed, image_masactrl

    def _panel(self, source_image, image_masactrl, prompt_src, prompt_tar):
        side = source_image.shape[-1]
        u8 = lambda t: (t.permute(1, 2, 0).detach().cpu().numpy() * 255).astype(np.uint8)
        instruct = txt_draw(f"source prompt: {prompt_src}\ntarget prompt: {prompt_tar}", target_size=(side, side))
        src = ((source_image[0].permute(1, 2, 0).detach().cpu().numpy() * 0.5 + 0.5) * 255).astype(np.uint8)
        return Image.fromarray(np.concatenate((np.array(instruct), src, u8(image_masactrl[0]), u8(image_masactrl[-1])), 1))

    def _images(self, image_path):
        source_image = load_image(image_path, self.model.device)
        image_gt = load_512(image_path)
        side = self._side()
        if side != 512:   # reduced test configurations only
            source_image = F.interpolate(source_image, (side, side))
            image_gt = np.array(Image.fromarray(image_gt).resize((side, side)))
        return source_image, image_gt

    def edit_image_directinversion_MasaCtrl(self, image_path, prompt_src, prompt_tar, guidance_scale, step=4, layper=10,
                                            return_stages=False):
        """run_editing_masactrl.py:87-133"""
        source_image, image_gt = self._images(image_path)
        inv = DirectInversion(model=self.model, num_ddim_steps=self.num_ddim_steps)
        _, _, x_stars, noise_loss_list = inv.invert(image_gt=image_gt, prompt=["", prompt_tar], guidance_scale=guidance_scale)
        x_t = x_stars[-1]
        _, image_masactrl = self._sample(prompt_tar, x_t.expand(2, -1, -1, -1), guidance_scale, step, layper, noise_loss_list)
        panel = self._panel(source_image, image_masactrl, prompt_src, prompt_tar)
        return (panel, dict(x_stars=x_stars, noise_loss_list=noise_loss_list, images=image_masactrl)) if return_stages else panel

    def edit_image_ddim_MasaCtrl(self, image_path, prompt_src, prompt_tar, guidance_scale, step=4, layper=10, return_stages=False):
        """run_editing_masactrl.py:135-175"""
        source_image, _ = self._images(image_path)
        start_code, latents_list = self.model.invert(source_image, "", guidance_scale=guidance_scale,
                                                     num_inference_steps=self.num_ddim_steps, return_intermediates=True)
        _, image_masactrl = self._sample(prompt_tar, start_code.expand(2, -1, -1, -1), guidance_scale, step, layper, None)
        panel = self._panel(source_image, image_masactrl, prompt_src, prompt_tar)
        return (panel, dict(x_stars=latents_list, images=image_masactrl)) if return_stages else panel


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rerun_exist_images", action="store_true")
    ap.add_argument("--data_path", type=str, default="data")
    ap.add_argument("--output_path", type=str, default="output")
    ap.add_argument("--edit_category_list", nargs="+", type=str, default=[str(i) for i in range(10)])
    ap.add_argument("--edit_method_list", nargs="+", type=str, default=["ddim+masactrl", "directinversion+masactrl"])
    ap.add_argument("--model_config", choices=("sd1", "small64"), default="sd1", help="small64: reduced-width test configuration")
    ap.add_argument("--num_ddim_steps", type=int, default=50)
    from pnpinversion_amd.checkpoint import add_weight_args, resolve_weights
    add_weight_args(ap)
    args = ap.parse_args(argv)
    from pnpinversion_amd.distributed import broadcast_weights, shard_items
    from pnpinversion_amd.config import SD1, SMALL64
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cfg = SD1 if args.model_config == "sd1" else SMALL64
    unet_sd, vae_sd, clip_sd, tokenizer = resolve_weights(args, cfg, rank)     # --checkpoint_dir | --synthetic_weights (loud)
    pipe = MasaCtrlPipeline(cfg, device="cuda:%d" % local_rank, text_encoder="native", tokenizer=tokenizer)
    if rank == 0:
        pipe.load_state_dict(unet_sd, vae_sd, clip_sd=clip_sd)
    if world > 1:
        broadcast_weights(pipe.engine, src=0)
    editor = MasaCtrlEditor(args.edit_method_list, torch.device("cuda", local_rank), num_ddim_steps=args.num_ddim_steps, pipeline=pipe)
    with open(os.path.join(args.data_path, "mapping_file.json")) as f:
        instructions = json.load(f)
    work = [(k, v) for k, v in instructions.items() if v["editing_type_id"] in args.edit_category_list]
    for key, item in shard_items(work, rank, world):
        src = item["original_prompt"].replace("[", "").replace("]", "")
        tgt = item["editing_prompt"].replace("[", "").replace("]", "")
        image_path = os.path.join(args.data_path, "annotation_images", item["image_path"])
        for method in args.edit_method_list:
            out_path = image_path.replace(args.data_path, os.path.join(args.output_path, method))
            if os.path.exists(out_path) and not args.rerun_exist_images:
                print(f"skip image [{image_path}] with [{method}]")
                continue
            print(f"editing image [{image_path}] with [{method}]")
            setup_seed()
            torch.cuda.empty_cache()
            edited = editor(method, image_path=image_path, prompt_src=src, prompt_tar=tgt, guidance_scale=7.5, step=4, layper=10)
            os.makedirs(os.path.dirname(out_path), exist_ok=True)
            edited.save(out_path)
            print("finish")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

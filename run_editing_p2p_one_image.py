#!/usr/bin/env python
"""Single-image driver with the CLI of the reference's run_editing_p2p_one_image.py, on the MI355X-native P2PEditor."""
import argparse

import torch

from pnpinversion_amd.checkpoint import add_weight_args, resolve_weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.p2p_editor import P2PEditor
from run_editing_p2p import setup_seed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_path", type=str, default="scripts/example_cake.jpg")
    ap.add_argument("--original_prompt", type=str, default="a round cake with orange frosting on a wooden plate")
    ap.add_argument("--editing_prompt", type=str, default="a square cake with orange frosting on a wooden plate")
    ap.add_argument("--blended_word", type=str, default="cake cake")
    ap.add_argument("--output_path", nargs="+", type=str, default=["directinversion+p2p.jpg"])
    ap.add_argument("--edit_method_list", nargs="+", type=str, default=["directinversion+p2p"])
    add_weight_args(ap)
    args = ap.parse_args()
    unet_sd, vae_sd, clip_sd, tokenizer = resolve_weights(args, SD1)           # --checkpoint_dir | --synthetic_weights (loud)
    from pnpinversion_amd.text import WordTokenizer
    editor = P2PEditor(args.edit_method_list, torch.device("cuda"), state_dicts=(unet_sd, vae_sd, clip_sd),
                       tokenizer=tokenizer or WordTokenizer())
    blended = args.blended_word.split(" ") if args.blended_word != "" else []
    for method, out_path in zip(args.edit_method_list, args.output_path):
        print(f"editing image [{args.image_path}] with [{method}]")
        setup_seed()
        edited = editor(method, image_path=args.image_path, prompt_src=args.original_prompt, prompt_tar=args.editing_prompt,
                        guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
                        blend_word=((blended[0],), (blended[1],)) if blended else None,
                        eq_params={"words": (blended[1],), "values": (2,)} if blended else None,
                        proximal="l0", quantile=0.75, use_inversion_guidance=True, recon_lr=1, recon_t=400)
        edited.save(out_path)
        print("finish")


if __name__ == "__main__":
    main()

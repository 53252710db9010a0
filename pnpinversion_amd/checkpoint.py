"""Loading a Stable-Diffusion-1.x checkpoint directory in the diffusers layout -- what the reference gets from
`StableDiffusionPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", ...)` (models/p2p_editor.py:23-24):

    <dir>/unet/diffusion_pytorch_model.{safetensors,bin}      UNet2DConditionModel state dict
    <dir>/vae/diffusion_pytorch_model.{safetensors,bin}       AutoencoderKL state dict
    <dir>/text_encoder/{model.safetensors,pytorch_model.bin}  transformers CLIPTextModel state dict
    <dir>/tokenizer/{vocab.json,merges.txt}                   CLIP byte-level BPE vocabulary -> text.ClipBPETokenizer

No checkpoint exists on the build / GPU boxes (no network), so the runs there use seeded synthetic weights; this module is the
path a user with a real checkpoint takes (`--checkpoint_dir` of the drivers, `P2PEditor(..., checkpoint_dir=...)`)."""
import os
import sys

import torch


def _load_sd(folder, names):
    for n in names:
        p = os.path.join(folder, n)
        if os.path.exists(p):
            if p.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError("none of %s under %s" % (", ".join(names), folder))


def load_checkpoint_dir(path):
    """-> (unet_sd, vae_sd, clip_sd, tokenizer)"""
    from .text import ClipBPETokenizer
    if not os.path.isdir(path):
        raise FileNotFoundError("checkpoint directory %r does not exist" % path)
    model_files = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.bin")
    unet = _load_sd(os.path.join(path, "unet"), model_files)
    vae = _load_sd(os.path.join(path, "vae"), model_files)
    clip = _load_sd(os.path.join(path, "text_encoder"), ("model.safetensors", "model.fp16.safetensors", "pytorch_model.bin"))
    clip = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in clip.items() if "position_ids" not in k}
    tok = ClipBPETokenizer(os.path.join(path, "tokenizer", "vocab.json"), os.path.join(path, "tokenizer", "merges.txt"))
    return unet, vae, clip, tok


SYNTHETIC_WARNING = ("WARNING: running on SEEDED SYNTHETIC (random) Stable-Diffusion weights and a word-level stand-in tokenizer -- the "
                     "saved panels are numerically meaningful (parity / timing) but are NOT edits of a trained model.  Pass "
                     "--checkpoint_dir <diffusers SD-1.x directory> for real weights.")


def add_weight_args(ap):
    ap.add_argument("--checkpoint_dir", type=str, default=None,
                    help="Stable-Diffusion-1.x checkpoint in the diffusers layout (unet/, vae/, text_encoder/, tokenizer/): what the "
                         "reference downloads as CompVis/stable-diffusion-v1-4")
    ap.add_argument("--synthetic_weights", action="store_true",
                    help="explicit opt-in: seeded random weights + word-level stand-in tokenizer (no checkpoint exists offline)")


def resolve_weights(args, cfg, rank=0):
    """-> (unet_sd, vae_sd, clip_sd, tokenizer or None).  Exactly one of --checkpoint_dir / --synthetic_weights must be given;
    state dicts are None on ranks other than `rank` 0 (they receive the packed arena by the start-up broadcast)."""
    if args.checkpoint_dir and args.synthetic_weights:
        raise SystemExit("--checkpoint_dir and --synthetic_weights are mutually exclusive")
    if args.checkpoint_dir:
        unet, vae, clip, tok = load_checkpoint_dir(args.checkpoint_dir)
        return (unet, vae, clip, tok) if rank == 0 else (None, None, None, tok)
    if not args.synthetic_weights:
        raise SystemExit("no weights: pass --checkpoint_dir <diffusers SD-1.x directory> (the reference loads CompVis/stable-diffusion-v1-4), "
                         "or --synthetic_weights to run on seeded random weights")
    if rank == 0:
        print(SYNTHETIC_WARNING, file=sys.stderr, flush=True)
        from . import weights
        return weights.unet_state_dict(cfg, 0), weights.vae_state_dict(cfg, 0), weights.clip_state_dict(cfg, 0), None
    return None, None, None, None

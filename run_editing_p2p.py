#!/usr/bin/env python
"""PIE-Bench sweep driver with the CLI of the reference's run_editing_p2p.py (same flags, same output tree, same
skip-if-exists resume), on the MI355X-native P2PEditor.  Under torch.distributed.run (one process per GPU) the ordered work
list is sharded round-robin over the ranks and the weight arena is broadcast once from rank 0 over RCCL / xGMI; there is no
per-image communication."""
import argparse
import json
import os
import random

import numpy as np
import torch
from PIL import Image

from pnpinversion_amd.checkpoint import add_weight_args, resolve_weights
from pnpinversion_amd.distributed import broadcast_weights, prepare_env, shard_items
from pnpinversion_amd.p2p_editor import P2PEditor

BATCHED_METHODS = ("directinversion+p2p",)     # methods with a several-images-per-launch entry point (always the lock-step schedule)


def mask_decode(encoded_mask, image_shape=(512, 512)):
    """PIE-Bench run-length mask: [start0, len0, start1, len1, ...] over the flattened image; the border is forced to 1
    (run_editing_p2p.py:11-27)."""
    n = image_shape[0] * image_shape[1]
    mask = np.zeros(n)
    runs = np.asarray(encoded_mask, dtype=np.int64).reshape(-1, 2)
    for start, length in runs:
        mask[start:start + min(length, n - start)] = 1
    mask = mask.reshape(image_shape)
    mask[0, :] = mask[-1, :] = 1
    mask[:, 0] = mask[:, -1] = 1
    return mask


def setup_seed(seed=1234):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rerun_exist_images", action="store_true")
    ap.add_argument("--data_path", type=str, default="data")
    ap.add_argument("--output_path", type=str, default="output")
    ap.add_argument("--edit_category_list", nargs="+", type=str, default=[str(i) for i in range(10)])
    ap.add_argument("--edit_method_list", nargs="+", type=str, default=["directinversion+p2p"])
    ap.add_argument("--batch_size", type=int, default=1, help="images per set of launches and GPU (not in the reference: it edits one by one)")
    ap.add_argument("--overlap_stages", dest="overlap_stages", action="store_true", default=True,
                    help="directinversion+p2p, batch_size 1 (default): invert the next image on a second HIP stream while this one is "
                         "edited -- same panels, about 12 %% more images per second")
    ap.add_argument("--no_overlap_stages", dest="overlap_stages", action="store_false", help="edit strictly one image after the other")
    ap.add_argument("--images_in_flight", type=int, default=3,
                    help="batch_size 1 (default 3; not in the reference, which edits one by one): image i is edited on library context "
                         "i %% N of this GPU, each context with its own HIP stream and worker thread -- the one-row launch chains of an edit "
                         "leave most of the chip idle between dependent launches, independent chains fill the gaps; same panels.  "
                         "Measured on MI355X: directinversion+p2p 0.90 -> 1.11 images/s at 3, null-text-inversion+p2p 2.1x at 4.  "
                         "1 = one image after the other (then --overlap_stages applies)")
    ap.add_argument("--model_config", choices=("sd1", "small64"), default="sd1", help="small64: reduced-width test configuration")
    ap.add_argument("--num_ddim_steps", type=int, default=50)
    add_weight_args(ap)
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    prepare_env()                      # dmabuf IPC + 127.0.0.1 rendezvous, before RCCL initialises
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from pnpinversion_amd.config import SD1, SMALL64
    from pnpinversion_amd.pipeline import NativePipeline
    cfg = SD1 if args.model_config == "sd1" else SMALL64
    unet_sd, vae_sd, clip_sd, tokenizer = resolve_weights(args, cfg, rank)     # --checkpoint_dir | --synthetic_weights (loud)
    batched = args.batch_size > 1 and any(m in BATCHED_METHODS for m in args.edit_method_list)
    if args.batch_size > 1:
        for m in args.edit_method_list:
            if m not in BATCHED_METHODS:
                print(f"WARNING: --batch_size {args.batch_size} applies to {BATCHED_METHODS} only; [{m}] is edited one image at a time")
    pipe = NativePipeline(cfg, device="cuda:%d" % local_rank, max_unet_rows=12 * (args.batch_size if batched else 1),
                          text_encoder="native", tokenizer=tokenizer)
    if rank == 0:
        pipe.load_state_dict(unet_sd, vae_sd, clip_sd=clip_sd)
    if world > 1:
        bstats = {}
        broadcast_weights(pipe.engine, src=0, stats=bstats)
        if rank == 0:
            print("weight arena broadcast over RCCL: %.1f MB to %d ranks in %.1f ms" % (bstats["bytes"] / 1e6, bstats["ranks"], bstats["ms"]))
    editor = P2PEditor(args.edit_method_list, torch.device("cuda", local_rank), num_ddim_steps=args.num_ddim_steps, pipeline=pipe)

    with open(os.path.join(args.data_path, "mapping_file.json")) as f:
        instructions = json.load(f)
    work = [(k, v) for k, v in instructions.items() if v["editing_type_id"] in args.edit_category_list]
    def fields(item):
        src = item["original_prompt"].replace("[", "").replace("]", "")
        tgt = item["editing_prompt"].replace("[", "").replace("]", "")
        image_path = os.path.join(args.data_path, "annotation_images", item["image_path"])
        blended = item["blended_word"].split(" ") if item["blended_word"] != "" else []
        return src, tgt, image_path, blended

    mine = list(shard_items(work, rank, world))
    for method in args.edit_method_list:
        todo = []
        for key, item in mine:
            src, tgt, image_path, blended = fields(item)
            _ = Image.fromarray(np.uint8(mask_decode(item["mask"])[:, :, None].repeat(3, 2))).convert("L")   # unused, as in the reference
            out_path = image_path.replace(args.data_path, os.path.join(args.output_path, method))
            if os.path.exists(out_path) and not args.rerun_exist_images:
                print(f"skip image [{image_path}] with [{method}]")
                continue
            todo.append((src, tgt, image_path, blended, out_path))
        nb = args.batch_size if method in BATCHED_METHODS else 1
        if args.images_in_flight > 1 and nb == 1 and len(todo) >= 2:      # nothing / one image to edit: no extra contexts
            setup_seed()
            stream_items = [(c[2], c[0], c[1], ((c[3][0],), (c[3][1],)) if c[3] else None,
                             {"words": (c[3][1],), "values": (2,)} if c[3] else None) for c in todo]
            panels = editor.edit_stream_in_flight(method, stream_items, n_flight=args.images_in_flight, guidance_scale=7.5, cross_replace_steps=0.4,
                                                  self_replace_steps=0.6, proximal="l0", quantile=0.75, use_inversion_guidance=True, recon_lr=1,
                                                  recon_t=400)
            for c, panel in zip(todo, panels):
                print(f"editing image [{c[2]}] with [{method}]")
                os.makedirs(os.path.dirname(c[4]), exist_ok=True)
                panel.save(c[4])
                print("finish")
            continue
        if args.overlap_stages and nb == 1 and method == "directinversion+p2p":
            setup_seed()
            stream_items = [(c[2], c[0], c[1], ((c[3][0],), (c[3][1],)) if c[3] else None,
                             {"words": (c[3][1],), "values": (2,)} if c[3] else None) for c in todo]
            for c, panel in zip(todo, editor.edit_stream_directinversion(stream_items, guidance_scale=7.5, cross_replace_steps=0.4,
                                                                         self_replace_steps=0.6)):
                print(f"editing image [{c[2]}] with [{method}]")
                os.makedirs(os.path.dirname(c[4]), exist_ok=True)
                panel.save(c[4])
                print("finish")
            continue
        for b0 in range(0, len(todo), nb):
            chunk = todo[b0:b0 + nb]
            for (_, _, image_path, _, _) in chunk:
                print(f"editing image [{image_path}] with [{method}]")
            setup_seed()
            if len(chunk) == 1:
                src, tgt, image_path, blended, _ = chunk[0]
                panels = [editor(method, image_path=image_path, prompt_src=src, prompt_tar=tgt, guidance_scale=7.5,
                                 cross_replace_steps=0.4, self_replace_steps=0.6,
                                 blend_word=((blended[0],), (blended[1],)) if blended else None,
                                 eq_params={"words": (blended[1],), "values": (2,)} if blended else None,
                                 proximal="l0", quantile=0.75, use_inversion_guidance=True, recon_lr=1, recon_t=400)]
            else:   # several images per set of launches (12 UNet rows each); same per-image schedule and results
                panels = editor.edit_images_directinversion(
                    [c[2] for c in chunk], [c[0] for c in chunk], [c[1] for c in chunk], guidance_scale=7.5,
                    cross_replace_steps=0.4, self_replace_steps=0.6,
                    blend_words=[((c[3][0],), (c[3][1],)) if c[3] else None for c in chunk],
                    eq_params=[{"words": (c[3][1],), "values": (2,)} if c[3] else None for c in chunk])
            for panel, c in zip(panels, chunk):
                os.makedirs(os.path.dirname(c[4]), exist_ok=True)
                panel.save(c[4])
                print("finish")
    if hasattr(editor, "close_peers"):
        editor.close_peers()             # the extra library contexts of the in-flight sweep
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""BASELINE config 3's launch shape and nothing else, for `rocprofv3 --kernel-trace --stats`: N images per set of launches (N-row
inversion forwards, 12 N-row lock-step forwards), the faithful directinversion+p2p schedule with the bench's prompts / controller.
usage: batched_only.py [N = 8] [ddim steps = 50] [repeats = 1]   -> one JSON line (ms per batch, images / s)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.p2p_editor import P2PEditor
from pnpinversion_amd.pipeline import NativePipeline
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pipe = NativePipeline(SD1, device="cuda:0", max_unet_rows=12 * nb, max_vae_images=2, text_encoder="native")
pipe.load_state_dict(weights.unet_state_dict(SD1, 0), weights.vae_state_dict(SD1, 0), clip_sd=weights.clip_state_dict(SD1, 0))
ed = P2PEditor(["directinversion+p2p"], "cuda:0", num_ddim_steps=steps, pipeline=pipe)
rng = np.random.RandomState(0)
imgs = [rng.randint(0, 256, (512, 512, 3)).astype(np.uint8) for _ in range(nb)]
src, tgt = "a cat sitting on a wooden chair", "a dog sitting on a wooden chair"
def run():
    return ed.edit_images_directinversion(imgs, [src] * nb, [tgt] * nb, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6,
                                          blend_words=[(("cat",), ("dog",))] * nb, eq_params=[{"words": ("dog",), "values": (2,)}] * nb)
run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(json.dumps({"images_per_launch_set": nb, "ddim_steps": steps, "ms_per_batch": dt * 1e3, "images_per_s": nb / dt}))

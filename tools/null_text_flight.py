"""Sweep throughput of one method string (METHOD, default null-text-inversion+p2p) against the number of images in flight (P2PEditor.edit_stream_in_flight), full width.
usage: null_text_flight.py [n_flight ...] (default 1 2 3 4)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd.p2p_editor import P2PEditor
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
M = os.environ.get("METHOD", "null-text-inversion+p2p")
ed = P2PEditor([M], "cuda", num_ddim_steps=int(os.environ.get("STEPS", "50")))
rng = np.random.RandomState(0)
def items(n):
    return [(rng.randint(0, 256, (512, 512, 3)).astype(np.uint8), "a cat sitting on a wooden chair", "a dog sitting on a wooden chair",
             (("cat",), ("dog",)), {"words": ("dog",), "values": (2,)}) for _ in range(n)]
kw = dict(guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6)
for n in ns:
    for _ in ed.edit_stream_in_flight(M, items(n), n_flight=n, **kw): pass        # builds / warms every context
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = int(os.environ.get("PER_LANE", "2")) * n
    for _ in ed.edit_stream_in_flight(M, items(k), n_flight=n, **kw): pass
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("n_flight %d: %d images in %.2f s -> %.2f s per image, %.3f images/s" % (n, k, dt, dt / k, k / dt), flush=True)
ed.close_peers()

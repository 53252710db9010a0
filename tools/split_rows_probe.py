"""Does a twelve-row lock-step forward run faster as independent row groups on separate HIP streams (the HBM-bound norms and launch floors
of one group under the GEMMs of the other -- what three images in flight gain across images)?  One engine per group, all on one packed
weight arena (pnpi_create_shared), one worker thread per engine, a join after every forward (the step kernels need all twelve rows).
usage: split_rows_probe.py -> gpurun_out/split_rows_probe.json"""
import json, os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
main = NativeEngine(SD1, max_unet_rows=12, max_vae_images=1)
main.load_state_dict({k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}, {k: v.cuda() for k, v in weights.vae_state_dict(SD1, 0).items()})
torch.cuda.synchronize()
out = {}
def bench_single(rows, n=12):
    lat = torch.randn(rows, 4, 64, 64, device="cuda"); ctx = torch.randn(rows, 77, 768, device="cuda")
    main.text_kv_precompute(ctx)
    for _ in range(3): main.unet(lat, 500, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): main.unet(lat, 500, None)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for r in (12, 8, 6, 4):
    out["single_%d" % r] = bench_single(r)
    print("one stream, %2d rows: %.3f ms" % (r, out["single_%d" % r]), flush=True)
def bench_split(groups, n=12):
    engs, streams = [], []
    for i, r in enumerate(groups):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            e = main if i == 0 and False else NativeEngine(SD1, max_unet_rows=max(r, 4), max_vae_images=1, share_weights_with=main)
            s.synchronize()
        engs.append(e); streams.append(s)
    lats = [torch.randn(r, 4, 64, 64, device="cuda") for r in groups]; ctxs = [torch.randn(r, 77, 768, device="cuda") for r in groups]
    torch.cuda.synchronize()
    for e, s, c in zip(engs, streams, ctxs):
        with torch.cuda.stream(s): e.text_kv_precompute(c)
    torch.cuda.synchronize()
    bar = threading.Barrier(len(groups) + 1)
    stop = []
    def worker(i):
        with torch.cuda.stream(streams[i]):
            while True:
                bar.wait()
                if stop: return
                engs[i].unet(lats[i], 500, None)
                streams[i].synchronize()
                bar.wait()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(groups))]
    for t in th: t.start()
    def step():
        bar.wait(); bar.wait()
    for _ in range(3): step()
    t0 = time.perf_counter()
    for _ in range(n): step()
    dt = (time.perf_counter() - t0) / n * 1e3
    stop.append(1); bar.wait()
    for t in th: t.join()
    for e in engs: e.close()
    return dt
for groups in ((6, 6), (4, 8), (4, 4, 4), (3, 3, 3, 3)):
    key = "split_" + "_".join(map(str, groups))
    out[key] = bench_split(groups)
    print("%d streams, rows %s (join after every forward): %.3f ms per twelve rows" % (len(groups), groups, out[key]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/split_rows_probe.json", "w"), indent=1)
main.close()

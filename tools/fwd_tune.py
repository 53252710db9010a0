"""In-forward tile tuning: every tile / split-K configuration forced on the whole UNet forward in turn (pnpi_set_tuning
igemm_force_cfg / igemm_force_split), per-launch HIP-event times collected from the profile dump and summed per layer shape.
Unlike tools/autotune2.py (one GEMM in a loop, operands warm in the 256 MB MALL) this times each shape with the epilogue it really
has (residual, GEGLU, GroupNorm statistics, V^T) and the cache state it really meets inside a forward.
usage: fwd_tune.py [rows ...] (default 12 1) -> gpurun_out/fwd_tune_b<rows>.json (the format tools/gen_tile_table.py reads)
FWD_TUNE_WORK=ctxgrad (rows = 1): the workload is one recording forward + reverse walk (pnpi_unet_context_grad, the null-text iteration), so
the dgrad shapes of the walk are tuned too -> gpurun_out/fwd_tune_b1bwd.json
FWD_TUNE_WORK=vae (rows = images per call, default 1 2): one VAE encode + one decode at 512 x 512 (the edit runs 1 + 5 of them per image)
-> gpurun_out/fwd_tune_b<rows>vae.json"""
import csv, json, os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
rows_list = [int(a) for a in sys.argv[1:]] or [12, 1]
NAME = {0: "128", 1: "64", 3: "256m", 4: "320", 5: "256n", 6: "256x320", 7: "256x256", 8: "64k4", 9: "128k2", 10: "128k2b", 11: "64k2", 12: "64x320", 13: "128n2", 14: "128b64", 15: "256nb64", 16: "pp256", 17: "pp320", 18: "128b64x3"}
SINGLE = [0, 1, 3, 4, 5, 12, 13, 14, 15, 10, 11, 9, 8, 16, 17, 18]
SPLIT_CFGS = [0, 1, 4, 5, 11, 14, 16, 17, 18]
SPLITS = [2, 3, 4, 6, 8, 12, 16]
VAE = os.environ.get("FWD_TUNE_WORK") == "vae"
if VAE and not sys.argv[1:]: rows_list = [1, 2]
eng = NativeEngine(SD1, max_unet_rows=4 if VAE else max(rows_list + [4]), max_vae_images=max(rows_list) if VAE else 1)
eng.load_state_dict({k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}, {k: v.cuda() for k, v in weights.vae_state_dict(SD1, 0).items()})
lib = eng.lib
def setk(**kw):
    for k, v in kw.items(): assert lib.pnpi_set_tuning(k.encode(), v) == 0, k
os.makedirs("gpurun_out", exist_ok=True)
dump = "gpurun_out/_fwd_tune_dump.csv"
os.environ["PNPI_PROFILE_DUMP"] = dump
NF = 3
BWD = os.environ.get("FWD_TUNE_WORK") == "ctxgrad"
if BWD: rows_list = [1]
for rows in rows_list:
    lat = torch.randn(rows, 4, 64, 64, device="cuda"); ctx = torch.randn(rows, 77, 768, device="cuda")
    d_eps = torch.randn(rows, 4, 64, 64, device="cuda") * 256
    work = (lambda: eng.unet_context_grad(lat, 500, ctx, d_eps)) if BWD else (lambda: eng.unet(lat, 500, None))
    if VAE:
        img = torch.rand(rows, 3, 512, 512, device="cuda") * 2 - 1; zlat = torch.randn(rows, 4, 64, 64, device="cuda")
        work = lambda: (eng.vae_encode(img), eng.vae_decode(zlat))
    else:
        eng.text_kv_precompute(ctx)
    combos = [(-1, 0)] + [(c, 0) for c in SINGLE] + [(c, s) for s in SPLITS for c in SPLIT_CFGS]
    if os.environ.get("FWD_TUNE_CFGS"):        # quick look at a few configurations: FWD_TUNE_CFGS="3,14"
        combos = [(-1, 0)] + [(int(c), 0) for c in os.environ["FWD_TUNE_CFGS"].split(",")]
        combos += [(int(c), int(s)) for c in os.environ["FWD_TUNE_CFGS"].split(",") for s in os.environ.get("FWD_TUNE_SPLITS", "").split(",") if s]
    us = collections.defaultdict(lambda: collections.defaultdict(float))    # shape -> "name/sN" -> us per forward (sum of its launches)
    cnt = collections.defaultdict(int)
    total = {}
    for cfg, split in combos:
        setk(igemm_force_cfg=cfg, igemm_force_split=split)
        work()
        eng.profile_begin()
        for _ in range(NF): work()
        eng.profile_end()
        per = collections.defaultdict(float); n = collections.defaultdict(int); tot = 0.0
        for r in csv.DictReader(open(dump)):
            if int(r["M"]) == 0 or int(r["cls"]) not in (0, 1, 2, 9): continue
            k = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["ksize"]), int(r["cfg"]), int(r["split"]))
            per[k] += float(r["us"]) / NF; n[k] += 1; tot += float(r["us"]) / NF
        total["auto" if cfg < 0 else "%s/s%d" % (NAME[cfg], max(split, 1))] = tot
        for (M, N, K, ks, c, s), t in per.items():
            if c not in NAME: continue
            sh = (M, N, K, ks)
            if cfg < 0:
                us[sh]["auto"] = us[sh].get("auto", 0.0) + t
                cnt[sh] += n[(M, N, K, ks, c, s)] // NF
                continue
            key = NAME[c] + ("/s%d" % s if s > 1 else "")
            # a forced configuration the launcher replaced (illegal split, no DMA path) shows up under what really ran; keep the minimum
            us[sh][key] = min(us[sh].get(key, 1e30), t)
    setk(igemm_force_cfg=-1, igemm_force_split=0)
    shapes = [{"M": M, "N": N, "K": K, "ks": ks, "launches": cnt[(M, N, K, ks)], "us": dict(v)} for (M, N, K, ks), v in sorted(us.items())]
    best = sum(min(v for k, v in s["us"].items() if k != "auto") for s in shapes)
    auto = sum(s["us"].get("auto", 0.0) for s in shapes)
    print("rows=%d GEMM us / forward: auto=%.0f per-shape best=%.0f  whole-forward forced:" % (rows, auto, best),
          " ".join("%s=%.0f" % kv for kv in sorted(total.items(), key=lambda kv: kv[1])[:8]), flush=True)
    for s in sorted(shapes, key=lambda s: -(s["us"].get("auto", 0) - min(v for k, v in s["us"].items() if k != "auto")))[:12]:
        b = min(((k, v) for k, v in s["us"].items() if k != "auto"), key=lambda kv: kv[1])
        print("  (%d,%d,%d,%d) x%d auto=%.0f best=%s %.0f" % (s["M"], s["N"], s["K"], s["ks"], s["launches"], s["us"].get("auto", 0), b[0], b[1]), flush=True)
    json.dump({"rows": rows, "per_forward": True, "shapes": shapes}, open("gpurun_out/fwd_tune_b%d%s.json" % (rows, "bwd" if BWD else ("vae" if VAE else "")), "w"), indent=1)

"""A/B of whole UNet forwards under the process-wide tuning knobs (pnpi_set_tuning), one engine, interleaved rounds.
usage: fwd_ab.py [rows ...]   (default 1 12) -> prints ms / forward per arm, writes gpurun_out/fwd_ab.json"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd import weights
from pnpinversion_amd.config import SD1
from pnpinversion_amd.engine import NativeEngine
rows_list = [int(a) for a in sys.argv[1:]] or [1, 12]
eng = NativeEngine(SD1, max_unet_rows=max(rows_list + [4]), max_vae_images=1)
eng.load_state_dict({k: v.cuda() for k, v in weights.unet_state_dict(SD1, 0).items()}, {k: v.cuda() for k, v in weights.vae_state_dict(SD1, 0).items()})
lib = eng.lib
DEFAULTS = {"igemm_wide": 1, "gn_inline_rows": 0, "igemm_force_cfg": -1, "igemm_v320": 1, "igemm_v256n": 1, "igemm_deep_rings": 1,
            "igemm_vt_lds": 1, "igemm_res_late": 0, "igemm_table": 1, "igemm_bias_init": 1, "igemm_sched": int(os.environ.get("FWD_AB_SCHED_DEFAULT", "0")), "tile_order": -1, "igemm_v128": 2, "attn_vt_perm": 1, "igemm_tapin": 0, "igemm_pp_only_n": 0, "attn_aug": 1, "gn_slab": 0}
ARMS = {"default": {}, "gn_slab": {"gn_slab": 1}, "no_attn_aug": {"attn_aug": 0}, "tapin": {"igemm_tapin": 1}, "no_pp": {"igemm_pp_only_n": -1}, "no_wide": {"igemm_wide": 0}, "gn_inline": {"gn_inline_rows": 1 << 20}, "no_deep_rings": {"igemm_deep_rings": 0},
        "no_vt_lds": {"igemm_vt_lds": 0}, "res_late": {"igemm_res_late": 1}, "no_table": {"igemm_table": 0}, "no_bias_init": {"igemm_bias_init": 0},
        "order0": {"tile_order": 0}, "order1": {"tile_order": 1}, "order2": {"tile_order": 2},
        "f0_v2": {"igemm_force_cfg": 0, "igemm_v128": 2, "attn_vt_perm": 1, "igemm_tapin": 0, "igemm_pp_only_n": 0, "attn_aug": 1}, "f0_v3": {"igemm_force_cfg": 0, "igemm_v128": 3}, "f0_v8": {"igemm_force_cfg": 0, "igemm_v128": 8},
        "f0_v1": {"igemm_force_cfg": 0, "igemm_v128": 1}, "f0_v0": {"igemm_force_cfg": 0, "igemm_v128": 0}, "f14": {"igemm_force_cfg": 14},
        "no_vt_perm": {"attn_vt_perm": 0},
        "sched": {"igemm_sched": 1}, "sched2": {"igemm_sched": 2}, "sched0": {"igemm_sched": 0}}
for kv in filter(None, os.environ.get("FWD_AB_DEFAULTS", "").split(",")):      # FWD_AB_DEFAULTS="gn_slab=1": what the default arm and the dump run with
    DEFAULTS[kv.split("=")[0]] = int(kv.split("=")[1])
if os.environ.get("FWD_AB_ARMS"): ARMS = {k: v for k, v in ARMS.items() if k in os.environ["FWD_AB_ARMS"].split(",")}
def setk(d):
    for k, v in {**DEFAULTS, **d}.items(): assert lib.pnpi_set_tuning(k.encode(), v) == 0, k
out = {}
for rows in rows_list:
    lat = torch.randn(rows, 4, 64, 64, device="cuda"); ctx = torch.randn(rows, 77, 768, device="cuda")
    n = 30 if rows <= 4 else 12
    res = {a: [] for a in ARMS}; res["kv_cached"] = []
    for rnd in range(3):
        for arm, kn in ARMS.items():
            setk(kn)
            for _ in range(2): eng.unet(lat, 500, ctx)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): eng.unet(lat, 500, ctx)
            torch.cuda.synchronize(); res[arm].append((time.perf_counter() - t0) / n * 1e3)
        setk({})
        eng.text_kv_precompute(ctx)
        for _ in range(2): eng.unet(lat, 500, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): eng.unet(lat, 500, None)
        torch.cuda.synchronize(); res["kv_cached"].append((time.perf_counter() - t0) / n * 1e3)
    out[rows] = {a: min(v) for a, v in res.items()}
    print("rows=%d ms/forward (min of 3 rounds):" % rows, " ".join("%s=%.3f" % (a, min(v)) for a, v in res.items()), flush=True)
    setk({})
    if os.environ.get("FWD_AB_DUMP"):
        os.environ["PNPI_PROFILE_DUMP"] = "gpurun_out/dump_r2_b%d.csv" % rows
        eng.text_kv_precompute(ctx)
        eng.profile_begin()
        for _ in range(3): eng.unet(lat, 500, None)
        cls = eng.profile_end()
        out[rows]["classes"] = {k: (v["launches"] // 3, round(v["total_ms"] / 3, 3)) for k, v in cls.items() if v["launches"]}
        print("  classes (launches, ms per forward):", out[rows]["classes"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/fwd_ab.json", "w"), indent=1)

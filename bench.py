#!/usr/bin/env python
"""bench.py -- edited images/sec of the direct-inversion + Prompt-to-Prompt hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one complete `P2PEditor("directinversion+p2p", ...)` edit of one 512x512 image per rank, FAITHFUL schedule of the
reference (models/p2p_editor.py:415-479): VAE encode, the discarded VAE decode, 50 B=1 DDIM-inversion UNet calls, 50 B=4
offset-calculation calls, 50 B=4 reconstruct-pass calls (AttentionStore), 50 B=4 edit-pass calls (AttentionRefine +
AttentionReweight + LocalBlend -- the PIE-Bench default controller, run_editing_p2p.py:120-138), 4 more VAE decodes, uint8
read-back and the 4-panel image: 650 UNet sample-forwards + 1 encode + 5 decodes = 535.9 TFLOP algorithmic.
Weights are seeded-synthetic SD-1.x (no checkpoint exists offline), input image and prompts are synthetic.

Multi-GPU: images are independent -> each rank edits its own image (weak scaling, no data-path collective); the only
collective is the start-up RCCL broadcast of the packed weight arena from rank 0 (untimed set-up).

Extra objects in the JSON line:
  roofline      dominant kernel class (the MFMA implicit-GEMM conv/linear kernel): algorithmic FLOPs / summed per-launch HIP-event
                durations, measured by bracketing every launch of one full edit with events on the library's stream.
  cpu_baseline  the CPU oracle (a port of the reference's fp32 PyTorch path, oracle/) timed on this host: one B=1 and one B=4
                full-width UNet forward + VAE encode + decode, scaled by the faithful schedule's call counts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_GFLOP = 803.27          # per sample-forward
TEXT_KV_GFLOP = 2.95         # of which: the 32 cross-attention to_k / to_v projections of the text context -- executed once
                             # per loop and context row by the text K/V cache instead of inside every forward
VAE_ENC_TFLOP, VAE_DEC_TFLOP = 1.117, 2.515
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16, MI355X_MICROARCH.md
PROMPT_SRC = "a cat sitting on a wooden chair"
PROMPT_TGT = "a dog sitting on a wooden chair"


def synthetic_image(seed):
    """Smooth seeded 512x512 RGB uint8 image (the reference's example JPEGs are not on the GPU box)."""
    g = np.random.Generator(np.random.Philox(key=[seed, 77]))
    low = g.uniform(0, 255, size=(16, 16, 3)).astype(np.float32)
    img = np.kron(low, np.ones((32, 32, 1), dtype=np.float32))
    img += g.normal(0, 6, size=img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def cpu_baseline(cfg, budget_s=40.0):
    """Oracle (CPU port of the reference's fp32 path) on the host cores: a BOUNDED sample (about 10-30 s of CPU work), scaled to
    one faithful edit by the schedule's call counts.  Thread count is capped: 200+ threads on these small convolutions is slower
    than 32."""
    from oracle import sd_oracle
    from pnpinversion_amd import weights
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    usd, vsd = weights.unet_state_dict(cfg, 0), weights.vae_state_dict(cfg, 0)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(4, 4, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = weights.synth_context(cfg, 4, seed=1)
    t, start = {}, time.perf_counter()
    with torch.no_grad():
        sd_oracle.unet_forward(usd, cfg, lat[:1, :, :16, :16].contiguous(), 500, ctx[:1])      # warm-up (thread pool, allocator)
        t0 = time.perf_counter(); sd_oracle.unet_forward(usd, cfg, lat[:1], 500, ctx[:1]); t["unet_b1"] = time.perf_counter() - t0
        if time.perf_counter() - start + 4 * t["unet_b1"] < budget_s:
            t0 = time.perf_counter(); sd_oracle.unet_forward(usd, cfg, lat, 500, ctx); t["unet_b4"] = time.perf_counter() - t0
            b4_note = "1x UNet B=4 (%.2fs)" % t["unet_b4"]
        else:
            t["unet_b4"] = 4 * t["unet_b1"]
            b4_note = "UNet B=4 taken as 4x B=1"
        # VAE at the full 512 x 512 (its attention block is quadratic in pixels: no scaling from a smaller size)
        S = cfg.sample_size * cfg.vae_scale
        img = torch.rand(1, 3, S, S, generator=g) * 2 - 1
        t0 = time.perf_counter(); sd_oracle.vae_encode_mean(vsd, cfg, img); t["vae_enc"] = time.perf_counter() - t0
        t0 = time.perf_counter(); sd_oracle.vae_decode(vsd, cfg, lat[:1]); t["vae_dec"] = time.perf_counter() - t0
    per_image = 50 * t["unet_b1"] + 150 * t["unet_b4"] + t["vae_enc"] + 5 * t["vae_dec"]
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "oracle (CPU port of the reference's fp32 path; the reference itself is not on this box) fp32, %d threads: 1x UNet B=1 (%.2fs), %s, VAE enc + dec at 512^2 (%.2fs, %.2fs); scaled x(50, 150, 1, 5) "
                      "= %.0f s/image; sample wall %.0fs" % (threads, t["unet_b1"], b4_note, t["vae_enc"], t["vae_dec"], per_image,
                                                             time.perf_counter() - start)}


def self_spawn(n):
    """Re-run this command as n ranks under torch.distributed.run (rendezvous on 127.0.0.1, a free port); returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the pruned-schedule and batched extras (short profiling runs)")
    ap.add_argument("--dry-run", action="store_true", help="process-launch / rendezvous plumbing only, on CPU with gloo (tests)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--batch-images", type=int, default=8,
                    help="after the headline measurement (one image at a time), also time this many images per set of launches "
                         "(configs[2] style sweep batching; reported under \"batched\", never as value); 0/1 = skip")
    ap.add_argument("--schedule", choices=("lockstep", "reference"), default="lockstep",
                    help="lockstep: offsets + reconstruction + edit passes share one 12-row UNet launch per timestep; "
                         "reference: the reference's phase order, one 4-row launch per pass and step (same work, same results)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU), exactly as the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` would
        sys.exit(self_spawn(args.gpus))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    dist = None
    if args.dry_run:
        # rendezvous / barrier / max-over-ranks / one-JSON-line plumbing only (CPU, gloo): tests/test_distributed_cpu.py
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"dry_run": True, "metric": "edited images/sec @ 512x512, 50 DDIM-inv + 50 denoise steps", "value": None,
                              "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "max_rank_seconds": float(dt.item())}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    from pnpinversion_amd.distributed import prepare_env
    prepare_env()                      # dmabuf IPC + 127.0.0.1 rendezvous, before RCCL initialises (also when the driver launches the ranks)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pnpinversion_amd.config import SD1
    from pnpinversion_amd.distributed import broadcast_weights
    from pnpinversion_amd.p2p_editor import P2PEditor
    from pnpinversion_amd.pipeline import NativePipeline
    from pnpinversion_amd import weights

    cfg = SD1
    pipe = NativePipeline(cfg, device="cuda:%d" % local_rank, max_unet_rows=max(12 * max(1, args.batch_images), 12) if args.schedule == "lockstep" else 4, max_vae_images=2,
                          text_encoder="native")        # prompts are embedded by the device CLIP text transformer (A1)
    if rank == 0:
        pipe.load_state_dict(weights.unet_state_dict(cfg, 0), weights.vae_state_dict(cfg, 0), clip_sd=weights.clip_state_dict(cfg, 0))
    bcast = {}
    if world > 1:
        broadcast_weights(pipe.engine, src=0, stats=bcast)       # the one collective: RCCL broadcast of the packed arena over xGMI
    editor = P2PEditor(["directinversion+p2p"], "cuda:%d" % local_rank, num_ddim_steps=args.ddim_steps, pipeline=pipe)
    editor.lockstep = args.schedule == "lockstep"
    eng = pipe.engine

    # synthetic inputs are generated before the timed region (the hot path starts at the decoded RGB image, as in the reference)
    n_pre = args.warmup + args.steps
    images = {i: synthetic_image(1000 * rank + i) for i in list(range(n_pre)) + [999]}

    def one_edit(i):
        img = images[i]
        return editor("directinversion+p2p", image_path=img, prompt_src=PROMPT_SRC, prompt_tar=PROMPT_TGT, guidance_scale=7.5,
                      cross_replace_steps=0.4, self_replace_steps=0.6, blend_word=(("cat",), ("dog",)),
                      eq_params={"words": ("dog",), "values": (2,)})

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_edit(i)
    # `clock` (rank 0): a fixed MFMA-only calibration kernel before and after the timed region (effective matrix-pipe GHz) and the part's
    # own sclk / socket power / temperature sampled from a side thread during it -- so that a swing of the headline between boxes or
    # rounds can be attributed from the record.  Both probes are OUTSIDE the timed region.
    from pnpinversion_amd.utils.gpu_clock import ClockSampler

    def mfma_probe():
        try:
            torch.cuda.synchronize()
            ghz, ms = eng.clock_probe()
            return {"effective_ghz": round(ghz, 4), "probe_ms": round(ms, 2)}
        except Exception as e:      # the probe must never take the headline line down
            return {"error": "%s: %s" % (type(e).__name__, e)}

    clock = {"mfma_probe_before": mfma_probe()} if rank == 0 else None
    sampler = ClockSampler(local_rank) if rank == 0 else None
    barrier()
    eng.reset_counters()
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    for i in range(args.steps):
        panel = one_edit(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if sampler is not None:
        sampler.__exit__(None, None, None)
        clock["during_timed_region"] = sampler.summary()
        clock["mfma_probe_after"] = mfma_probe()
        clock["note"] = ("mfma_probe: v_mfma_f32_32x32x16_f16 back to back on every SIMD (pnpi_clock_probe, pseudo-random operands, HIP events): "
                         "matrix-pipe cycles / elapsed; nominal 2.4 GHz.  during_timed_region: the part's own sensors, side thread")
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ctr = eng.counters()
    assert panel.size == (2048, 512)

    def executed_flops(c):
        """algorithmic FLOPs actually executed: forwards that read the text K/V cache do not project the context"""
        fw, cached = c["unet_sample_forwards"], c["unet_sample_forwards_cached_kv"]
        return ((fw - cached) * UNET_GFLOP + cached * (UNET_GFLOP - TEXT_KV_GFLOP) + c["text_kv_rows"] * TEXT_KV_GFLOP) * 1e9 + \
            c["vae_encodes"] * VAE_ENC_TFLOP * 1e12 + c["vae_decodes"] * VAE_DEC_TFLOP * 1e12

    def per_image_flops_faithful():
        return executed_flops(ctr) / max(1, args.steps)

    # per-kernel-class roofline: one more full edit with every launch bracketed by HIP events (rank 0, outside the timed region)
    roofline, classes = None, None
    if rank == 0:
        import csv, tempfile
        dump = os.path.join(tempfile.gettempdir(), "pnpi_bench_launches_%d.csv" % os.getpid())
        os.environ["PNPI_PROFILE_DUMP"] = dump          # one record per launch: class, shape, HIP-event time, kernel template
        eng.profile_begin()
        one_edit(999)
        classes = eng.profile_end()
        os.environ.pop("PNPI_PROFILE_DUMP", None)
        gemm = {k: classes[k] for k in ("igemm128", "igemm64", "igemm64_splitk", "igemm_wide")}
        tot_ms = sum(v["total_ms"] for v in gemm.values())
        tot_fl = sum(v["flops"] for v in gemm.values())
        # the dominant KERNEL: launches grouped by the igemm_dma_kernel template instance they ran (the name a rocprofv3 kernel trace
        # shows), split-K launches included (their bracket also holds the reduce launch: the rate is slightly under-stated)
        kern = {}
        if os.path.exists(dump):
            for r in csv.DictReader(open(dump)):
                if r["kernel"].startswith("igemm"):
                    k = kern.setdefault(r["kernel"].replace(" ", ","), {"launches": 0, "us": 0.0, "flops": 0.0, "bytes": 0.0})
                    k["launches"] += 1; k["us"] += float(r["us"]); k["flops"] += float(r["flops"]); k["bytes"] += float(r["bytes"])
            os.remove(dump)
        if not kern:      # no per-launch records (unwritable temp dir): fall back to the dominant kernel CLASS
            cdom = max(gemm, key=lambda k: gemm[k]["total_ms"])
            kern = {"igemm class %s" % cdom: {"launches": gemm[cdom]["launches"], "us": gemm[cdom]["total_ms"] * 1e3,
                                              "flops": gemm[cdom]["flops"], "bytes": gemm[cdom]["bytes"]}}
        dom = max(kern, key=lambda k: kern[k]["us"])
        d = kern[dom]
        ach = d["flops"] / (d["us"] * 1e-6) / 1e12
        roofline = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
                    "alg_flop_per_launch": d["flops"] / d["launches"], "alg_bytes_per_launch": d["bytes"] / d["launches"],
                    "launches": d["launches"], "avg_launch_us": d["us"] / d["launches"],
                    "share_of_igemm_time": d["us"] * 1e-3 / tot_ms,
                    "all_igemm_achieved": tot_fl / (tot_ms * 1e-3) / 1e12,
                    "kernels": {k: {"launches": v["launches"], "ms": round(v["us"] * 1e-3, 3), "tflops": round(v["flops"] / (v["us"] * 1e-6) / 1e12, 1)}
                                for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["us"])[:8]},
                    "classes": {k: {"launches": v["launches"], "ms": round(v["total_ms"], 3),
                                    "tflops": (v["flops"] / (v["total_ms"] * 1e-3) / 1e12) if v["total_ms"] > 0 and v["flops"] > 0 else None,
                                    "GBps": (v["bytes"] / (v["total_ms"] * 1e-3) / 1e9) if v["total_ms"] > 0 and v["bytes"] > 0 else None}
                                for k, v in classes.items()}}

    # the two loops of one more edit timed on their own (rank 0, outside the timed region): the 50 one-row DDIM-inversion forwards and the
    # 50 twelve-row lock-step steps of the dual-branch loop north_star's 40 % target is stated on (models/p2p/p2p_guidance_forward.py:135-173)
    def loop_phases(run, n_img):
        """`run()` once more with a device synchronisation around pnpi_ddim_invert and pnpi_direct_edit (host wall clock): the n_img-row
        inversion forwards and the 12 n_img-row lock-step steps of the dual-branch loop timed on their own"""
        marks = {}

        def timed(name, fn):
            def wrapper(*a, **k):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                out_ = fn(*a, **k)
                torch.cuda.synchronize()
                marks[name] = marks.get(name, 0.0) + time.perf_counter() - t_
                return out_
            return wrapper

        orig_inv, orig_edit = eng.ddim_invert, eng.direct_edit
        eng.ddim_invert, eng.direct_edit = timed("invert", orig_inv), timed("lockstep", orig_edit)
        try:
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            run()
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t_all
        finally:
            eng.ddim_invert, eng.direct_edit = orig_inv, orig_edit
        if "invert" not in marks or "lockstep" not in marks:
            return None
        n = args.ddim_steps
        rows = 12 * n_img
        loop_flop = n * rows * (UNET_GFLOP - TEXT_KV_GFLOP) * 1e9 + rows * TEXT_KV_GFLOP * 1e9      # text K / V projected once per loop
        return {"ddim_inversion_ms": marks["invert"] * 1e3, "inversion_rows": n_img, "inversion_forward_ms": marks["invert"] * 1e3 / n,
                "lockstep_loop_ms": marks["lockstep"] * 1e3, "lockstep_rows": rows, "lockstep_step_ms": marks["lockstep"] * 1e3 / n,
                "lockstep_loop_tflops": loop_flop / marks["lockstep"] / 1e12,
                "lockstep_loop_mfma_frac": loop_flop / marks["lockstep"] / 1e12 / MFMA_PEAK_TFLOPS,
                "rest_ms": (t_all - marks["invert"] - marks["lockstep"]) * 1e3,
                "note": "one more run with a device synchronisation around pnpi_ddim_invert and pnpi_direct_edit (host wall clock); "
                        "rest = VAE encode / decodes, text encoder, controller tables, panel"}

    # the two loops of one more edit timed on their own (rank 0, outside the timed region): the 50 one-row DDIM-inversion forwards and the
    # 50 twelve-row lock-step steps of the dual-branch loop north_star's 40 % target is stated on (models/p2p/p2p_guidance_forward.py:135-173)
    phases = None
    if rank == 0 and args.schedule == "lockstep":
        phases = loop_phases(lambda: one_edit(999), 1)
        if phases is not None:      # the names round 4's line used
            phases["one_row_forward_ms"] = phases["inversion_forward_ms"]
            phases["twelve_row_step_ms"] = phases["lockstep_step_ms"]

    # extra (never `value`): the pruned-equivalent schedule of SURVEY Note D, FLOPs from the library's counters
    pruned = None
    if args.schedule == "lockstep" and not args.no_extras:
        editor.schedule = "pruned"
        one_edit(999)
        barrier()
        eng.reset_counters()
        tp = time.perf_counter()
        one_edit(999)
        barrier()
        dtp = time.perf_counter() - tp
        cp = eng.counters()
        editor.schedule = "faithful"
        if dist is not None:
            tt = torch.tensor([dtp], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtp = float(tt.item())
        pruned = {"value": world / dtp, "unit": "images/s", "ms_per_image": dtp * 1e3, "unet_sample_forwards_per_image": cp["unet_sample_forwards"],
                  "executed_tflop_per_image": executed_flops(cp) / 1e12, "executed_tflops_per_gpu": executed_flops(cp) / dtp / 1e12,
                  "note": "same edit, source latent assigned from the inversion trajectory (3-row launches); parity-tested against the faithful schedule"}

    # extra (never `value`): BASELINE config 4 -- "null-text-inversion+p2p" on the same image: DDIM inversion, the per-step optimisation of
    # the unconditional embedding (up to 10 Adam iterations, each a recording UNet forward + the reverse walk to the 77 x 768 embedding),
    # then the two guidance passes with the per-step embeddings.  FLOPs are the library's executed counters (backward launches included).
    null_text = None
    if world == 1 and not args.no_extras:
        try:
            ed_nt = P2PEditor(["null-text-inversion+p2p"], "cuda:%d" % local_rank, num_ddim_steps=args.ddim_steps, pipeline=pipe)
            barrier()
            eng.reset_counters()
            tn = time.perf_counter()
            ed_nt("null-text-inversion+p2p", image_path=images[999], prompt_src=PROMPT_SRC, prompt_tar=PROMPT_TGT, guidance_scale=7.5,
                  cross_replace_steps=0.4, self_replace_steps=0.6, blend_word=(("cat",), ("dog",)), eq_params={"words": ("dog",), "values": (2,)})
            barrier()
            dtn = time.perf_counter() - tn
            cn = eng.counters()
            ex = cn["executed_gemm_flops"] + cn["executed_attn_flops"]
            null_text = {"value": 1.0 / dtn, "unit": "images/s", "s_per_image": dtn, "unet_sample_forwards": cn["unet_sample_forwards"],
                         "unet_backward_rows": cn["unet_backward_rows"], "executed_tflop_per_image": ex / 1e12, "executed_tflops_per_gpu": ex / dtn / 1e12,
                         "note": "null-text-inversion+p2p, %d DDIM steps x 10 Adam iterations (synthetic weights never reach the early-stop "
                                 "threshold); executed FLOPs = 2*M*N*K of every GEMM / attention launch incl. the backward pass" % args.ddim_steps}
            # the same method with four images in flight (P2PEditor.edit_stream_in_flight: image i on library context i % 4, each with its
            # own HIP stream and worker thread): the null-text path is one-row launch chains almost throughout, independent chains fill the gaps
            try:
                kw2 = dict(guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6)
                its = lambda lo, n: [(synthetic_image(7000 + lo + j), PROMPT_SRC, PROMPT_TGT, (("cat",), ("dog",)), {"words": ("dog",), "values": (2,)})
                                     for j in range(n)]
                nf = 4
                for _ in ed_nt.edit_stream_in_flight("null-text-inversion+p2p", its(0, nf), n_flight=nf, **kw2):     # builds and warms the contexts
                    pass
                barrier()
                t2 = time.perf_counter()
                n2 = 2 * nf
                for _ in ed_nt.edit_stream_in_flight("null-text-inversion+p2p", its(nf, n2), n_flight=nf, **kw2):
                    pass
                barrier()
                dt2 = time.perf_counter() - t2
                null_text["in_flight"] = {"n_flight": nf, "value": n2 / dt2, "unit": "images/s", "s_per_image": dt2 / n2, "images": n2,
                                          "note": "same edits, image i on library context i % n_flight (own HIP stream and worker thread each)"}
                ed_nt.close_peers()
            except Exception as e:
                null_text["in_flight"] = {"error": "%s: %s" % (type(e).__name__, e)}
        except Exception as e:   # an extra must never take the headline line down
            null_text = {"error": "%s: %s" % (type(e).__name__, e)}

    batched = None
    if args.batch_images > 1 and args.schedule == "lockstep" and not args.no_extras:
        nb = args.batch_images
        batch_images = {i: [synthetic_image(5000 + 1000 * rank + nb * i + j) for j in range(nb)] for i in (0, 1)}

        def batch_edit(i):
            imgs = batch_images[i]
            return editor.edit_images_directinversion(imgs, [PROMPT_SRC] * nb, [PROMPT_TGT] * nb, guidance_scale=7.5,
                                                      cross_replace_steps=0.4, self_replace_steps=0.6,
                                                      blend_words=[(("cat",), ("dog",))] * nb,
                                                      eq_params=[{"words": ("dog",), "values": (2,)}] * nb)
        batch_edit(0)
        barrier()
        eng.reset_counters()
        tb = time.perf_counter()
        batch_edit(1)
        barrier()
        dtb = time.perf_counter() - tb
        cb = eng.counters()             # the batched run's OWN executed FLOPs (text K / V rows and cached forwards differ from the one-image run)
        if dist is not None:
            tt = torch.tensor([dtb], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtb = float(tt.item())
        batched = {"images_per_launch_set_per_gpu": nb, "value": nb * world / dtb, "unit": "images/s", "ms_per_batch": dtb * 1e3,
                   "whole_path_mfma_frac": executed_flops(cb) / dtb / 1e12 / MFMA_PEAK_TFLOPS,
                   "unet_sample_forwards_per_image": cb["unet_sample_forwards"] / nb,
                   "note": "BASELINE config 3's launch shape (batch = %d per GPU): same faithful schedule per image; %d-row inversion launches, "
                           "%d-row lock-step launches" % (nb, nb, 12 * nb)}
        # the same batches with the NEXT batch's nb-row inversion on a second context / HIP stream under this batch's 12 nb-row loop
        # (P2PEditor.edit_stream_images_directinversion: same kernels, same panels) -- the overlap the one-image path has in `pipelined`
        try:
            mkb = lambda k: ([synthetic_image(20000 + 1000 * rank + nb * k + j) for j in range(nb)], [PROMPT_SRC] * nb, [PROMPT_TGT] * nb,
                             [(("cat",), ("dog",))] * nb, [{"words": ("dog",), "values": (2,)}] * nb)
            kwb = dict(guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6)
            for _ in editor.edit_stream_images_directinversion([mkb(0)], **kwb):      # builds and warms the second context
                pass
            barrier()
            n_b = 3
            tq = time.perf_counter()
            for _ in editor.edit_stream_images_directinversion([mkb(1 + k) for k in range(n_b)], **kwb):
                pass
            barrier()
            dtq = time.perf_counter() - tq
            if dist is not None:
                tt = torch.tensor([dtq], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtq = float(tt.item())
            batched["pipelined"] = {"value": n_b * nb * world / dtq, "unit": "images/s", "batches": n_b, "ms_per_batch": dtq / n_b * 1e3,
                                    "note": "the next batch's %d-row inversion on a second HIP stream under this batch's %d-row lock-step loop "
                                            "(the first batch's inversion is not overlapped and is inside the timed region)" % (nb, 12 * nb)}
            editor.close_peers()
        except Exception as e:   # an extra must never take the headline line down
            batched["pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            # the same loop north_star's 40 % target is stated on, at config 3's batch: 50 steps of one 12 nb-row launch set
            try:
                batched["phases"] = loop_phases(lambda: batch_edit(1), nb)
            except Exception as e:   # an extra must never take the headline line down
                batched["phases"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # extra (never `value`): sweep throughput with the NEXT image's inversion on a second context / HIP stream under this image's
    # lock-step loop (P2PEditor.edit_stream_directinversion); same kernels, same panels
    pipelined = None
    if args.schedule == "lockstep" and not args.no_extras:
        try:
            def items(lo, n):
                return [(synthetic_image(9000 + 1000 * rank + lo + j), PROMPT_SRC, PROMPT_TGT, (("cat",), ("dog",)), {"words": ("dog",), "values": (2,)})
                        for j in range(n)]
            n_pipe = 6
            warm, work = items(0, 2), items(2, n_pipe)
            for _ in editor.edit_stream_directinversion(warm, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6):
                pass
            barrier()
            tq = time.perf_counter()
            for _ in editor.edit_stream_directinversion(work, guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6):
                pass
            barrier()
            dtq = time.perf_counter() - tq
            if dist is not None:
                tt = torch.tensor([dtq], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtq = float(tt.item())
            pipelined = {"value": n_pipe * world / dtq, "unit": "images/s", "images": n_pipe, "ms_per_image": dtq / n_pipe * 1e3,
                         "note": "faithful schedule, one image per lock-step loop; the next image's one-row inversion overlapped on a second HIP stream "
                                 "(the first image's inversion is not overlapped and is inside the timed region)"}
        except Exception as e:   # an extra must never take the headline line down
            pipelined = {"error": "%s: %s" % (type(e).__name__, e)}

    # extra (never `value`): the headline method with three images in flight (P2PEditor.edit_stream_in_flight: image i on library context
    # i % 3 of this GPU, own HIP stream and worker thread each; same kernels, same panels) -- the sweep mode of run_editing_p2p.py
    in_flight = None
    if args.schedule == "lockstep" and not args.no_extras:
        try:
            nf = 3
            mk = lambda lo, n: [(synthetic_image(12000 + 1000 * rank + lo + j), PROMPT_SRC, PROMPT_TGT, (("cat",), ("dog",)), {"words": ("dog",), "values": (2,)})
                                for j in range(n)]
            kwf = dict(guidance_scale=7.5, cross_replace_steps=0.4, self_replace_steps=0.6)
            for _ in editor.edit_stream_in_flight("directinversion+p2p", mk(0, nf), n_flight=nf, **kwf):
                pass
            barrier()
            tf = time.perf_counter()
            n_if = 4 * nf
            for _ in editor.edit_stream_in_flight("directinversion+p2p", mk(nf, n_if), n_flight=nf, **kwf):
                pass
            barrier()
            dtf = time.perf_counter() - tf
            if dist is not None:
                tt = torch.tensor([dtf], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtf = float(tt.item())
            in_flight = {"n_flight": nf, "value": n_if * world / dtf, "unit": "images/s", "images": n_if, "ms_per_image": dtf / n_if * 1e3,
                         "note": "faithful schedule per image; image i on library context i % n_flight (own HIP stream and worker thread each)"}
            editor.close_peers()
        except Exception as e:   # an extra must never take the headline line down
            in_flight = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        # HBM-side traffic / MFMA busy of the dominant kernel per launch: FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES from separate
        # rocprofv3 --pmc passes of THIS command with --ddim-steps 2 (tools/profile_round.sh: the counter service dies on the ~100 000
        # launches of a full run; same kernels, same launch shapes, same 1 : 1 mix of one-row and twelve-row forwards), reduced by
        # tools/pmc_summary.py and committed under profiles/ with the source hash of the kernels they were measured on.  A summary made
        # from other kernel sources than the running tree is NOT reported (traffic: null + the reason).
        from pnpinversion_amd.build import source_hash
        prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        sha = source_hash()
        if roofline is not None:
            roofline["source_sha16"] = sha
            tag = roofline["kernel"].replace(" ", "")           # igemm_pp_kernel<192,320,1,4,0>
            rounds = ("round6", "round5", "round4")
            cand = [os.path.join(prof, "%s_pmc_traffic_bench.json" % r) for r in rounds]
            have = [(c_, json.load(open(c_))) for c_ in cand if os.path.exists(c_)]
            match = [(c_, j_) for c_, j_ in have if j_.get("source_sha16") == sha]
            if not have:
                roofline["traffic_note"] = "no committed counter summary (profiles/round6_pmc_traffic_bench.json)"
            elif not match:
                roofline["traffic_note"] = "%s was measured on kernel sources %s, this tree is %s: not reported" % (
                    os.path.relpath(have[0][0], os.path.dirname(prof)), have[0][1].get("source_sha16"), sha)
            else:
                pmc, src_json = match[0]
                rel_pmc = os.path.relpath(pmc, os.path.dirname(prof))
                for name, v in src_json["kernels"].items():
                    if v.get("template") == tag:
                        roofline["traffic"] = v["traffic_bytes"]
                        roofline["traffic_over_alg"] = v["traffic_bytes"] / roofline["alg_bytes_per_launch"]
                        if "mfma_util" in v:
                            roofline["mfma_busy"] = v["mfma_util"]
                        roofline["traffic_source"] = "%s (same kernel sources, %s): %s" % (rel_pmc, sha, src_json.get("source"))
                        break
                else:
                    roofline["traffic_note"] = "the dominant kernel %s has no entry in %s" % (tag, rel_pmc)
            # the rocprofv3 --kernel-trace --stats average of the same kernel under this command (committed summary + its source hash): the
            # HIP-event bracket above also carries the packet-processing gap in front of the kernel, which differs between boxes of the pool
            for r in rounds:
                stats_csv, meta = os.path.join(prof, "%s_bench_kernel_stats.csv" % r), os.path.join(prof, "%s_bench_kernel_stats.meta.json" % r)
                if not (os.path.exists(stats_csv) and os.path.exists(meta) and json.load(open(meta)).get("source_sha16") == sha):
                    continue
                import csv
                name, targs = tag.split("<")
                want = "%d%sI" % (len(name), name) + "".join("Li%sE" % a for a in targs.rstrip(">").split(",")) + "E"
                for row in csv.DictReader(open(stats_csv)):
                    if want in row["Name"] or tag in row["Name"].replace(" ", ""):       # mangled, or demangled "void igemm_pp_kernel<192, 320, 1, 4, 0>(GemmP)"
                        us = float(row["AverageNs"]) / 1e3
                        roofline["rocprof"] = {"avg_launch_us": us, "calls": int(row["Calls"]),
                                               "achieved": roofline["alg_flop_per_launch"] / us / 1e6,
                                               "frac": roofline["alg_flop_per_launch"] / us / 1e6 / MFMA_PEAK_TFLOPS,
                                               "source": "profiles/%s_bench_kernel_stats.csv: rocprofv3 --kernel-trace --stats of `bench.py --steps 2 "
                                                         "--warmup 1 --no-extras --no-cpu-baseline` (faithful edits only; same kernel sources, %s)" % (r, sha)}
                        break
                break
        # the same fractions against what the matrix pipes of THIS box sustain (information only; `frac` stays against the nominal peak):
        # nominal peak x (MFMA-only probe clock / 2.4 GHz nominal)
        try:
            ghz = [clock[k]["effective_ghz"] for k in ("mfma_probe_before", "mfma_probe_after") if "effective_ghz" in clock.get(k, {})]
            if ghz and roofline is not None:
                sustained = MFMA_PEAK_TFLOPS * (sum(ghz) / len(ghz)) / 2.4
                roofline["peak_at_mfma_probe_clock"] = sustained
                roofline["frac_of_peak_at_mfma_probe_clock"] = roofline["achieved"] / sustained
                if phases is not None:
                    phases["lockstep_loop_frac_of_peak_at_mfma_probe_clock"] = phases["lockstep_loop_tflops"] / sustained
        except Exception:
            pass
        n_img = args.steps * world
        per_rank_flops = executed_flops(ctr)
        out = {
            "metric": "edited images/sec @ 512x512, 50 DDIM-inv + 50 denoise steps",
            "value": n_img / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (fp32 accumulate; latents / scheduler math fp32)", "data": "synthetic",
            "config": {"workload": "single 512x512 image per rank, SD-1.x (seeded synthetic weights), directinversion+p2p, "
                                   "faithful schedule: 650 UNet sample-forwards + 1 VAE encode + 5 VAE decodes per image, "
                                   "Refine+Reweight+LocalBlend controller; text K/V projected once per loop (not per forward)",
                       "ddim_steps": args.ddim_steps, "images_per_step_per_gpu": 1, "schedule": args.schedule,
                       "unet_sample_forwards_per_image": ctr["unet_sample_forwards"] / max(1, args.steps),
                       "algorithmic_tflop_per_image": per_rank_flops / max(1, args.steps) / 1e12},
            "rccl_ranks": world if world > 1 else 0,
            "bcast_ms": bcast.get("ms"), "bcast_mb": (bcast.get("bytes", 0) / 1e6) if bcast else None,   # the start-up weight broadcast (untimed set-up)
            "whole_path_tflops_per_gpu": per_rank_flops / dt / 1e12,
            "whole_path_mfma_frac": per_rank_flops / dt / 1e12 / MFMA_PEAK_TFLOPS,
            "roofline": roofline, "phases": phases, "clock": clock,
        }
        if pruned is not None:
            out["pruned_schedule"] = pruned
        if batched is not None:
            out["batched"] = batched
        if pipelined is not None:
            out["pipelined"] = pipelined
        if in_flight is not None:
            out["in_flight"] = in_flight
        if null_text is not None:
            out["null_text"] = null_text
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (bounded sample, see cpu_baseline)
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

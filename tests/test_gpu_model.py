"""Whole-graph parity of the native UNet / VAE against the CPU oracle (oracle/sd_oracle.py) on identical seeded weights.
Stated tolerances (fp16 activations / fp32 accumulation vs fp32 oracle), cf. SURVEY.md 8(d):
  UNet single forward: rel-L2(eps) <= 4e-3;  VAE encode mean: rel-L2 <= 4e-3;  VAE decode: rel-L2 <= 6e-3, |pixel diff| mean <= 2/255."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd_oracle  # noqa: E402  (checker only)
from pnpinversion_amd import weights  # noqa: E402
from pnpinversion_amd.config import SD1, TINY16, SMALL64  # noqa: E402
from pnpinversion_amd.engine import NativeEngine  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _lat(cfg, rows, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(rows, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=g)


@pytest.fixture(scope="module")
def tiny():
    cfg = TINY16
    usd, vsd = weights.unet_state_dict(cfg, 1), weights.vae_state_dict(cfg, 1)
    eng = NativeEngine(cfg, max_unet_rows=8, max_vae_images=2)
    eng.load_state_dict(usd, vsd)
    n, names = eng.missing_weights()
    assert n == 0, names[:10]
    yield cfg, usd, vsd, eng
    eng.close()


@pytest.mark.parametrize("rows,t", [(1, 981), (4, 500), (8, 1)])
def test_unet_tiny(tiny, rows, t):
    cfg, usd, vsd, eng = tiny
    lat = _lat(cfg, rows, 5)
    ctx = weights.synth_context(cfg, rows, seed=7)
    got = eng.unet(lat, t, ctx, rows_per_image=1)
    with torch.no_grad():
        ref = sd_oracle.unet_forward(usd, cfg, lat, t, ctx)
    assert torch.isfinite(got).all()
    assert rel(got, ref) < 4e-3, rel(got, ref)


def test_text_kv_cache_and_timestep_bias_table_are_exact(tiny):
    """pnpi_text_kv_precompute + pnpi_unet_forward(context=NULL), and the per-timestep (conv1 bias + time embedding) table:
    the same GEMM / GEMV launches moved out of the forward, so the result is bit-identical to the in-forward evaluation."""
    cfg, usd, vsd, eng = tiny
    lat = _lat(cfg, 4, 41)
    ctx = weights.synth_context(cfg, 4, seed=42)
    lib = eng.lib
    lib.pnpi_set_tuning(b"temb_cache", 0)
    ref = eng.unet(lat, 777, ctx).clone()
    lib.pnpi_set_tuning(b"temb_cache", 1)
    first, again = eng.unet(lat, 777, ctx).clone(), eng.unet(lat, 777, ctx).clone()      # fills the table row, then reads it
    assert torch.equal(first, ref) and torch.equal(again, ref)
    eng.text_kv_precompute(ctx)
    c0 = eng.counters()
    cached = eng.unet(lat, 777, None)
    c1 = eng.counters()
    assert torch.equal(cached, ref)
    assert c1["unet_sample_forwards_cached_kv"] - c0["unet_sample_forwards_cached_kv"] == 4 and c0["text_kv_rows"] >= 4
    with pytest.raises(Exception, match="pnpi_text_kv_precompute"):
        eng.unet(lat[:2], 777, None)                                                     # cache holds 4 rows
    # a weight (re)load invalidates both caches
    eng.load_state_dict(usd, None)
    with pytest.raises(Exception, match="pnpi_text_kv_precompute"):
        eng.unet(lat, 777, None)
    assert torch.equal(eng.unet(lat, 777, ctx), ref)
    # level-2 loops: cached (default) vs projecting the context inside every forward
    from pnpinversion_amd.p2p.scheduler_dev import DDIMSchedulerDev
    sch = DDIMSchedulerDev(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)
    sch.bind(eng)
    sch.set_timesteps(3)
    ts = sch.timesteps.numpy()
    a = eng.ddim_invert(lat[:1], ctx[2:3], ts).clone()
    nl_a, lat_a = eng.direct_edit(a, ctx[None], [None], ts, 7.5)          # 8 rows: offsets + one guidance pass
    lib.pnpi_set_tuning(b"text_kv", 0)
    try:
        b = eng.ddim_invert(lat[:1], ctx[2:3], ts)
        nl_b, lat_b = eng.direct_edit(b, ctx[None], [None], ts, 7.5)
    finally:
        lib.pnpi_set_tuning(b"text_kv", 1)
    assert torch.equal(a, b) and torch.equal(nl_a, nl_b) and torch.equal(lat_a, lat_b)


def test_attention_callback_fallback_for_controllers_without_descriptor(tiny):
    """SURVEY 8b level 1: a controller object of the reference's protocol (`controller(attn, is_cross, place)`, called at each of the
    32 attention sites on the materialised [B*heads, N, M] probabilities -- attention_control.py:20-47,178-190) that the library has no
    kernel descriptor for runs through pnpi_set_attention_callback.  Checked: (1) an identity controller reproduces the fused forward,
    (2) the oracle's EditController driven through the call-back path matches both the CPU oracle and the fused descriptor path,
    (3) an arbitrary user controller (not expressible as a descriptor) matches the oracle run with the same hook, (4) exceptions raised
    in the controller surface in Python."""
    from oracle import p2p_oracle as po
    from pnpinversion_amd.p2p import attention_control as ac
    from pnpinversion_amd.pipeline import NativePipeline
    from pnpinversion_amd.text import WordTokenizer
    from types import SimpleNamespace
    cfg, usd, vsd, eng = tiny
    lat = _lat(cfg, 4, 51)
    lat[1], lat[3] = lat[0], lat[2]
    ctx = weights.synth_context(cfg, 4, seed=52)
    fused = eng.unet(lat, 500, ctx).clone()
    seen = []

    def identity(attn, is_cross, place, layer):
        seen.append((tuple(attn.shape), is_cross, place, layer))

    eng.set_attention_callback(identity, rows=4)
    via_cb = eng.unet(lat, 500, ctx).clone()
    eng.set_attention_callback(None)
    assert len(seen) == 32 and [s[3] for s in seen] == list(range(32)) and [s[1] for s in seen] == [False, True] * 16
    assert seen[0][0] == (4 * cfg.heads, 256, 256) and seen[1][0] == (4 * cfg.heads, 256, 77) and seen[12][2] == "mid"
    assert rel(via_cb, fused) < 3e-3, rel(via_cb, fused)          # P rounded to fp16 at a different point than in the flash kernel
    assert torch.equal(eng.unet(lat, 500, ctx), fused)            # the fused path is back

    # (2) the oracle's edit controller as the callback vs the native descriptor for the same tables, first step of a 3-step schedule
    prompts = ["a round cake with orange frosting on a wooden plate", "a square cake with orange frosting on a wooden plate"]
    tok = WordTokenizer()
    steps = 3
    c = ac.make_controller(SimpleNamespace(tokenizer=tok), prompts, True, {"default_": 0.4}, 0.6, None, {"words": ("square",), "values": (2,)},
                           num_ddim_steps=steps)
    tables = {"kind": "replace", "mapper": c.prev_controller.mapper[0], "equalizer": c.equalizer.reshape(-1),
              "cross_alpha": c.cross_replace_alpha.reshape(steps + 1, 77), "self_range": c.num_self_replace, "lb": None}
    dev_tables = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in tables.items()}
    hook_gpu = po.EditController(32, dev_tables)
    eng.set_attention_callback(lambda attn, is_cross, place, layer: hook_gpu(attn, is_cross, place), rows=4)
    cb_edit = eng.unet(lat, 500, ctx).clone()
    eng.set_attention_callback(None)
    native_edit = eng.unet(lat, 500, ctx, rows_per_image=4, ctrls=[c.tables()], cur_step=0)
    with torch.no_grad():
        ref_edit = sd_oracle.unet_forward(usd, cfg, lat, 500, ctx, po.EditController(32, tables))
    assert rel(cb_edit, ref_edit) < 4e-3, rel(cb_edit, ref_edit)
    assert rel(cb_edit, native_edit) < 4e-3, rel(cb_edit, native_edit)
    assert rel(fused[3], ref_edit[3]) > 3 * rel(cb_edit[3], ref_edit[3])          # the edit is visible in the target row

    # (3) + (4) through the drop-in surface: register_attention_control(model, <object without .tables()>)
    class HalveFirstKeys:                 # a user controller no descriptor can express
        def __init__(self): self.calls = 0
        def __call__(self, attn, is_cross, place):
            self.calls += 1
            if is_cross and place == "up":
                attn = attn.clone()
                attn[:, :, 1:4] *= 0.5
            return attn

    pipe = NativePipeline.__new__(NativePipeline)
    from pnpinversion_amd.pipeline import NativeUNet
    unet = NativeUNet(eng)
    user = HalveFirstKeys()
    ac.register_attention_control(SimpleNamespace(unet=unet), user)
    got = unet(lat, 500, encoder_hidden_states=ctx)["sample"]
    assert user.calls == 32
    with torch.no_grad():
        want = sd_oracle.unet_forward(usd, cfg, lat, 500, ctx, HalveFirstKeys())
    assert rel(got, want) < 4e-3, rel(got, want)
    ac.register_attention_control(SimpleNamespace(unet=unet), None)
    assert torch.equal(unet(lat, 500, encoder_hidden_states=ctx)["sample"], fused)

    class Boom:
        def __call__(self, attn, is_cross, place): raise KeyError("boom")
    ac.register_attention_control(SimpleNamespace(unet=unet), Boom())
    with pytest.raises(KeyError, match="boom"):
        unet(lat, 500, encoder_hidden_states=ctx)
    ac.register_attention_control(SimpleNamespace(unet=unet), None)
    assert torch.equal(unet(lat, 500, encoder_hidden_states=ctx)["sample"], fused)
    with pytest.raises(TypeError, match="no kernel descriptor"):
        ac.register_attention_control(SimpleNamespace(unet=unet), object())
        unet(lat, 500, encoder_hidden_states=ctx)
    ac.register_attention_control(SimpleNamespace(unet=unet), None)


def test_vae_tiny(tiny):
    cfg, usd, vsd, eng = tiny
    g = torch.Generator().manual_seed(9)
    f = cfg.vae_scale
    S = cfg.sample_size * f
    img = torch.rand(2, 3, S, S, generator=g) * 2 - 1
    with torch.no_grad():
        ref_m = sd_oracle.vae_encode_mean(vsd, cfg, img)
    got_m = eng.vae_encode(img)
    assert rel(got_m, ref_m) < 4e-3, rel(got_m, ref_m)
    z = torch.randn(2, 4, cfg.sample_size, cfg.sample_size, generator=g)
    with torch.no_grad():
        ref_d = sd_oracle.vae_decode(vsd, cfg, z)
    got_d = eng.vae_decode(z)
    assert rel(got_d, ref_d) < 6e-3, rel(got_d, ref_d)
    # uint8 image paths (utils/utils.py:58-80)
    u8 = (torch.rand(1, S, S, 3, generator=g) * 255).to(torch.uint8)
    x = u8[0].float() / 127.5 - 1
    with torch.no_grad():
        ref_z = sd_oracle.vae_encode_mean(vsd, cfg, x.permute(2, 0, 1)[None]) * 0.18215
    got_z = eng.image2latent(u8)
    assert rel(got_z, ref_z) < 4e-3
    with torch.no_grad():
        xr = sd_oracle.vae_decode(vsd, cfg, 1 / 0.18215 * ref_z)
    ref_img = ((xr / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)
    got_img = eng.latent2image(ref_z).cpu().numpy()
    assert np.abs(got_img.astype(np.int32) - ref_img.astype(np.int32)).mean() < 2.0


def test_unet_and_vae_sd1_full_width():
    """Full SD-1.x width (859.5 M-parameter UNet, 83.7 M VAE): one B=1 UNet forward, VAE at 256x256."""
    cfg = SD1
    usd = weights.unet_state_dict(cfg, 0)
    vsd = weights.vae_state_dict(cfg, 0)
    eng = NativeEngine(cfg, max_unet_rows=4, max_vae_images=1)
    eng.load_state_dict(usd, vsd)
    assert eng.missing_weights()[0] == 0
    lat = _lat(cfg, 1, 11)
    ctx = weights.synth_context(cfg, 1, seed=12)
    got = eng.unet(lat, 481, ctx)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        ref = sd_oracle.unet_forward(usd, cfg, lat, 481, ctx)
    assert rel(got, ref) < 4e-3, rel(got, ref)
    # B=4 rows must equal 4 independent B=1 evaluations (batch invariance of the native kernels)
    lat4 = torch.cat([lat, _lat(cfg, 3, 13)])
    ctx4 = torch.cat([ctx, weights.synth_context(cfg, 3, seed=14)])
    got4 = eng.unet(lat4, 481, ctx4)
    assert rel(got4[:1], ref) < 4e-3
    g = torch.Generator().manual_seed(15)
    img = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        ref_m = sd_oracle.vae_encode_mean(vsd, cfg, img)
        z = torch.randn(1, 4, 32, 32, generator=g)
        ref_d = sd_oracle.vae_decode(vsd, cfg, z)
    assert rel(eng.vae_encode(img), ref_m) < 4e-3
    assert rel(eng.vae_decode(z), ref_d) < 6e-3
    eng.close()


def test_weight_arena_broadcast_over_rccl_world1():
    """The one collective of the path (RCCL broadcast of the packed weight arena, distributed.py) on a real device buffer:
    a world-size-1 NCCL group is all a 1-GPU box allows, but it exercises the zero-copy arena view, the chunking and the RCCL
    call itself; the world-size-2 logic runs on gloo in tests/test_distributed_cpu.py."""
    import torch.distributed as dist
    from pnpinversion_amd.config import TINY16
    from pnpinversion_amd.distributed import arena_tensor, broadcast_weights
    from pnpinversion_amd.engine import NativeEngine
    from pnpinversion_amd import weights
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = NativeEngine(TINY16, max_unet_rows=2, max_vae_images=1)
        eng.load_state_dict(weights.unet_state_dict(TINY16, 3), weights.vae_state_dict(TINY16, 3))
        lat = torch.randn(2, 4, 16, 16, device="cuda")
        ctx = torch.randn(2, 77, TINY16.cross_dim, device="cuda")
        before = eng.unet(lat, 300, ctx).clone()
        t = arena_tensor(eng)
        assert t.is_cuda and t.dtype == torch.uint8 and t.numel() == eng.weight_arena()[1] and t.data_ptr() == eng.weight_arena()[0]
        snap = t.clone()
        n = broadcast_weights(eng, src=0)
        assert n == t.numel() and torch.equal(t, snap)
        assert torch.equal(eng.unet(lat, 300, ctx), before)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,cfg", [("tiny", TINY16), ("sd1", SD1)])
def test_clip_text_encoder_against_transformers_golden(name, cfg):
    """pnpi_text_encode (A1: model.text_encoder(ids)[0], inversion.py:290-306) vs transformers' CLIPTextModel on the same seeded
    weights (tests/golden/clip_*.npz): rel-L2 <= 4e-3 (fp16 activations, fp32 accumulation / softmax / LayerNorm statistics)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_%s.npz" % name))
    eng = NativeEngine(cfg, max_unet_rows=1, max_vae_images=0)
    ids = torch.from_numpy(g["input_ids"])
    with pytest.raises(Exception, match="weights not loaded: clip"):
        eng.text_encode(ids)                                   # fails loudly until the clip.* weights are there
    eng.load_state_dict(clip_sd=weights.clip_state_dict(cfg, int(g["seed"])))
    out = eng.text_encode(ids)
    ref = torch.from_numpy(g["hidden"])
    r = ((out.cpu() - ref).norm() / ref.norm()).item()
    assert out.shape == ref.shape and r < 4e-3, r
    # causal: the embedding at position p depends on tokens <= p only
    ids2 = ids.clone()
    ids2[:, 40:] = 1234
    out2 = eng.text_encode(ids2)
    assert torch.equal(out2[:, :40], out[:, :40]) and not torch.equal(out2[:, 40:], out[:, 40:])
    eng.close()


def test_pipeline_with_native_text_encoder():
    """NativePipeline(text_encoder="native"): init_prompt's embeddings come from the device CLIP transformer end to end."""
    from pnpinversion_amd.pipeline import NativePipeline, NativeTextEncoder
    from pnpinversion_amd.p2p.inversion import DirectInversion
    pipe = NativePipeline.synthetic(TINY16, seed=1, max_unet_rows=4, max_vae_images=1, text_encoder="native")
    assert isinstance(pipe.text_encoder, NativeTextEncoder)
    inv = DirectInversion(pipe, num_ddim_steps=2)
    inv.init_prompt(["a cat on a chair", "a dog on a chair"])
    assert inv.context.shape == (4, 77, TINY16.cross_dim) and inv.context.is_cuda
    assert torch.equal(inv.context[0], inv.context[1]) and not torch.equal(inv.context[2], inv.context[3])
    ref = sd_oracle.clip_text_forward(weights.clip_state_dict(TINY16, 1), TINY16,
                                      pipe.tokenizer(["a cat on a chair"], padding="max_length", max_length=77).input_ids)
    assert ((inv.context[2].cpu() - ref[0]).norm() / ref[0].norm()).item() < 4e-3
    pipe.engine.close()


def test_attention_store_keep_maps_and_npi_slerp(tiny):
    """Opt-in AttentionStore(keep_maps=True) (attention_control.py:214-248, what visualisation callers read): the <= 32^2-token
    conditional-half maps of two UNet calls, summed, against the oracle's StoreController; and utils.slerp_tensor (negative-prompt
    inversion's npi_interp, utils/utils.py:7-25) against its formula in float64."""
    from oracle import p2p_oracle as po
    from pnpinversion_amd.p2p import attention_control as ac
    from pnpinversion_amd.pipeline import NativeUNet
    from pnpinversion_amd.utils.utils import slerp_tensor
    from types import SimpleNamespace
    cfg, usd, vsd, eng = tiny
    lat = _lat(cfg, 4, 61)
    ctx = weights.synth_context(cfg, 4, seed=62)
    unet = NativeUNet(eng)
    store = ac.AttentionStore(keep_maps=True)
    ac.register_attention_control(SimpleNamespace(unet=unet), store)
    assert ac.is_callback_controller(unet.controller)
    ref = po.StoreController(unet.num_att_layers)
    for t in (800, 300):
        got = unet(lat, t, encoder_hidden_states=ctx)["sample"]
        with torch.no_grad():
            want = sd_oracle.unet_forward(usd, cfg, lat, t, ctx, ref)
        assert rel(got, want) < 4e-3
    assert store.cur_step == 2 == ref.cur_step
    avg = store.get_average_attention()
    assert set(avg) == set(ref.attention_store)
    n = 0
    for key in avg:
        assert len(avg[key]) == len(ref.attention_store[key]), key
        for a, b in zip(avg[key], ref.attention_store[key]):
            assert a.shape == b.shape and rel(a, b / 2) < 5e-3, (key, rel(a, b / 2))
            n += 1
    assert n == 32                                                        # TINY16: every site has <= 32^2 tokens
    # the default AttentionStore stays the declarative no-op (flash kernels, nothing stored)
    plain = ac.AttentionStore()
    ac.register_attention_control(SimpleNamespace(unet=unet), plain)
    assert not ac.is_callback_controller(unet.controller) and plain.tables() is None
    unet(lat, 800, encoder_hidden_states=ctx)
    assert plain.attention_store == {} and plain.cur_step == 1
    ac.register_attention_control(SimpleNamespace(unet=unet), None)
    # slerp
    g = torch.Generator().manual_seed(5)
    lo, hi = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    got = slerp_tensor(0.3, lo, hi)
    l64, h64 = lo.double().flatten(1), hi.double().flatten(1)
    om = torch.acos(((l64 / l64.norm(dim=1, keepdim=True)) * (h64 / h64.norm(dim=1, keepdim=True))).sum(1))
    want = (torch.sin(0.7 * om) / torch.sin(om)).unsqueeze(1) * l64 + (torch.sin(0.3 * om) / torch.sin(om)).unsqueeze(1) * h64
    assert rel(got, want.reshape(lo.shape)) < 1e-6


def test_shared_weight_arena_contexts():
    """pnpi_create_shared: a further context borrows the parent's packed weights -- same arena pointer (no second copy), ready without a
    load, bit-identical forward on its own stream, and it refuses to load weights itself."""
    import ctypes as C
    from pnpinversion_amd import _capi
    from pnpinversion_amd.config import SMALL64
    from pnpinversion_amd.engine import NativeEngine
    cfg = SMALL64
    main = NativeEngine(cfg, max_unet_rows=4, max_vae_images=1)
    usd, vsd = weights.unet_state_dict(cfg, 3), weights.vae_state_dict(cfg, 3)
    main.load_state_dict(usd, vsd)
    free_before = torch.cuda.mem_get_info()[0]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        child = NativeEngine(cfg, max_unet_rows=4, max_vae_images=1, share_weights_with=main)
    pa, pb, na, nb = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
    assert main.lib.pnpi_weight_arena(main.h, C.byref(pa), C.byref(na)) == 0 and main.lib.pnpi_weight_arena(child.h, C.byref(pb), C.byref(nb)) == 0
    assert pa.value == pb.value and na.value == nb.value
    assert child.missing_weights()[0] == 0
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(2, 4, cfg.sample_size, cfg.sample_size, generator=g).cuda()
    ctx = weights.synth_context(cfg, 2, seed=6).cuda()
    a = main.unet(lat, 481, ctx)
    with torch.cuda.stream(side):
        b = child.unet(lat, 481, ctx)
        side.synchronize()
    assert torch.equal(a, b)
    with pytest.raises(_capi.PnpiError, match="borrows its weights"):
        child.load_state_dict(usd, vsd)
    with pytest.raises(_capi.PnpiError):
        NativeEngine(cfg, max_unet_rows=4, max_vae_images=1, share_weights_with=child)      # share from the owner, not from a borrower
    child.close()
    main.close()

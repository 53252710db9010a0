"""Per-kernel parity: every hand-written HIP kernel against a plain PyTorch fp32 reference of the same op, through the C ABI.
Tolerances: inputs are fp16-representable, accumulation is fp32, outputs are rounded to fp16 once ->
rel-L2 <= 2e-3 (GEMM-like), <= 3e-3 (attention: P is rounded to fp16 before the second MFMA)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.gpu_util import Ctx, ptr, rel_err, max_err  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ctx():
    c = Ctx()
    yield c
    c.close()


def h16(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().to(DEV)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,cfg,split", [
    (256, 256, 128, 0, 0), (256, 256, 128, 1, 0), (256, 256, 512, 2, 4),
    (130, 70, 72, -1, 0), (4096, 320, 320, -1, 0), (308, 640, 768, -1, 0), (64, 1280, 5120, -1, 0),
    (1, 8, 8, -1, 0), (33, 4, 2880, 1, 0), (100, 36, 1032, 2, 3),
    (512, 256, 128, 3, 0), (700, 320, 640, 3, 0),        # 256x128 tile (ragged M and N)
    (4096, 320, 320, 4, 0), (700, 640, 640, 4, 0), (300, 200, 128, 4, 0), (130, 328, 64, 4, 0),     # 128x320 tile, ragged M / N
    (512, 320, 2048, 4, 4), (1000, 1280, 960, 4, 2),                                                 # 128x320 tile, split-K
    (256, 256, 512, 5, 0), (1000, 768, 320, 5, 0), (77, 520, 192, 5, 0), (512, 512, 1024, 5, 2),     # 128x256 tile
    (1000, 640, 640, 6, 0), (300, 330, 128, 6, 0), (700, 512, 256, 7, 0),                            # 8-wave 256x320 / 256x256 tiles
    (256, 320, 2880, 8, 0), (130, 70, 200, 8, 0), (64, 1280, 11520, 8, 3), (300, 192, 640, 11, 0),   # 64x64 tile, 4 / 2 k-groups of waves
    (768, 1280, 1280, 9, 0), (200, 130, 96, 9, 0), (512, 256, 4096, 10, 2), (128, 128, 64, 10, 0),   # 128x128 tile, 2 k-groups
    (1000, 640, 640, 12, 0), (100, 300, 128, 12, 0), (512, 320, 2048, 12, 2), (700, 384, 320, 13, 0),  # 64x320 tile; 128x128 two-stage
    (700, 384, 320, 14, 0), (512, 256, 4096, 14, 3), (1000, 768, 320, 15, 0), (77, 520, 192, 15, 0),   # 128-byte-row variants of 128x128 / 128x256
    # 8-wave ping-pong kernel (igemm_pp.inc): 16 = 256x256, 17 = 192x320; K-tile counts 1, 2, 3 (prologue / drain paths), ragged M / N, split-K
    (256, 256, 64, 16, 0), (256, 256, 128, 16, 0), (1000, 640, 640, 16, 0), (300, 328, 128, 16, 0), (700, 512, 256, 16, 0), (1000, 1280, 960, 16, 2),
    (4096, 1536, 320, 16, 0), (384, 640, 192, 17, 0), (1000, 640, 640, 17, 0), (130, 328, 64, 17, 0), (192, 320, 2880, 17, 0), (4096, 320, 320, 17, 0),
    (512, 320, 2048, 17, 4), (768, 1280, 11520, 17, 16), (3072, 1280, 1280, 17, 0),
    (40000, 320, 128, 17, 0), (33000, 256, 64, 16, 0),          # a linear layer is one image row of M pixels: M beyond 2^15
    (3072, 1280, 1280, 18, 0), (700, 384, 320, 18, 0), (512, 256, 4096, 18, 3), (130, 70, 72, 18, 0),   # 128x128, 128-byte rows, three stages
])
def test_gemm(ctx, M, N, K, cfg, split):
    a = h16(M, K, seed=1)
    w = h16(N, K, scale=1.0 / math.sqrt(K), seed=2)
    bias = torch.randn(N, device=DEV)
    res = h16(M, N, seed=3)
    ldo = (N + 3) // 4 * 4
    out = torch.zeros(M, ldo, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 0.5, ptr(bias), ptr(res), ptr(out), ldo, 1 << 30, None, 0, 0, 1, cfg, split)
    ref = 0.5 * (a.float() @ w.float().t()) + bias + res.float()
    assert rel_err(out[:, :N], ref) < 2e-3, (rel_err(out[:, :N], ref), max_err(out[:, :N], ref))


@pytest.mark.parametrize("key,val,cfg", [("igemm_v320", 0, 4), ("igemm_v320", 2, 4), ("igemm_v256n", 0, 5), ("igemm_v256n", 2, 5)])
def test_gemm_wide_tile_variants(ctx, key, val, cfg):
    """Non-default variants of the wide tiles (default 1 = two-stage ring of 64-byte rows with the two-pass LDS epilogue):
    0 = three stages / whole-tile epilogue, 2 = 128-byte rows."""
    M, N, K = 900, 640, 704
    a, w = h16(M, K, seed=21), h16(N, K, scale=1.0 / math.sqrt(K), seed=22)
    bias, res = torch.randn(N, device=DEV), h16(M, N, seed=23)
    out = torch.zeros(M, N, dtype=torch.half, device=DEV)
    lib = ctx.lib
    assert lib.pnpi_set_tuning(key.encode(), val) == 0
    try:
        ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, ptr(bias), ptr(res), ptr(out), N, 1 << 30, None, 0, 0, 1, cfg, 0)
        torch.cuda.synchronize()
    finally:
        assert lib.pnpi_set_tuning(key.encode(), 1) == 0
    ref = a.float() @ w.float().t() + bias + res.float()
    assert rel_err(out, ref) < 2e-3, rel_err(out, ref)
    assert lib.pnpi_set_tuning(b"no_such_knob", 1) != 0


@pytest.mark.parametrize("cfg,N,v320", [(0, 320, 0), (1, 192, 0), (4, 320, 0), (4, 640, 1), (5, 512, 0), (5, 256, 1), (6, 320, 0), (7, 256, 0),
                                        (8, 192, 1), (9, 256, 1), (12, 320, 1), (13, 256, 1), (14, 256, 1), (15, 512, 1), (16, 256, 1), (16, 512, 1),
                                        (17, 320, 1), (17, 640, 1)])
def test_conv_epilogue_groupnorm_statistics(ctx, cfg, N, v320):
    """The per-(m-tile, channel) sum / sum-of-squares partials a conv epilogue hands to the consumer GroupNorm: fp32 sums of the
    STORED fp16 values, every tile configuration (incl. the two-pass epilogue of the 2-stage wide tiles), ragged last m-tile."""
    B, Cin, H = 3, 64, 20                       # M = 1200 rows: ragged for 64- and 128-row tiles
    x = h16(B, Cin, H, H, seed=31)
    w = h16(N, Cin, 3, 3, scale=1.0 / math.sqrt(9 * Cin), seed=32)
    bias = torch.randn(N, device=DEV)
    out = torch.zeros(B, H, H, N, dtype=torch.half, device=DEV)
    M = B * H * H
    stats = torch.full(((M + 63) // 64, N, 2), float("nan"), device=DEV)
    rows = C.c_int(0)
    xn, wp = nhwc(x), pack_w(w)
    for key in (b"igemm_v320", b"igemm_v256n"):
        assert ctx.lib.pnpi_set_tuning(key, v320) == 0
    try:
        ctx.call("pnpi_op_conv_stats", ptr(xn), None, Cin, 0, B, H, H, 3, 1, 1, 0, H, H, ptr(wp), ptr(bias), None, N, ptr(out), cfg, 0,
                 ptr(stats), C.byref(rows))
        torch.cuda.synchronize()
    finally:
        for key in (b"igemm_v320", b"igemm_v256n"):
            ctx.lib.pnpi_set_tuning(key, 1)
    tr = rows.value
    assert tr == {1: 64, 8: 64, 11: 64, 12: 64, 6: 256, 7: 256, 16: 64, 17: 64}.get(cfg, 128)    # the ping-pong kernel: 64-row sub-blocks
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 2e-3
    o = out.reshape(M, N).float()
    nt = (M + tr - 1) // tr
    for t in range(nt):
        blk = o[t * tr:(t + 1) * tr]
        assert torch.allclose(stats[t, :, 0], blk.sum(0), rtol=1e-5, atol=1e-3), t
        assert torch.allclose(stats[t, :, 1], (blk * blk).sum(0), rtol=1e-5, atol=1e-3), t
    # bit-reproducible: a second launch gives the same bits (fixed summation order, no atomics)
    stats2 = torch.zeros_like(stats)
    ctx.call("pnpi_op_conv_stats", ptr(xn), None, Cin, 0, B, H, H, 3, 1, 1, 0, H, H, ptr(wp), ptr(bias), None, N, ptr(out), cfg, 0,
             ptr(stats2), C.byref(rows))
    torch.cuda.synchronize()
    if v320 == 1:
        assert torch.equal(stats[:nt], stats2[:nt])


def test_gemm_transposed_region(ctx):
    # QKV-style: columns [0, 128) plain, columns [128, 192) written transposed per batch item (V^T), tokens = 40 per item
    Bn, T, K, N, col0 = 3, 40, 64, 192, 128
    M = Bn * T
    a = h16(M, K, seed=4)
    w = h16(N, K, scale=0.1, seed=5)
    out = torch.zeros(M, col0, dtype=torch.half, device=DEV)
    vt = torch.zeros(Bn, N - col0, T, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, None, None, ptr(out), col0, col0, ptr(vt), T, 0, T, -1, 0)
    ref = a.float() @ w.float().t()
    assert rel_err(out, ref[:, :col0]) < 2e-3
    ref_vt = ref[:, col0:].reshape(Bn, T, N - col0).permute(0, 2, 1)
    assert rel_err(vt, ref_vt) < 2e-3
    # fp32 NCHW output (conv_out style): everything transposed
    o32 = torch.zeros(Bn, N, T, dtype=torch.float32, device=DEV)
    ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, None, None, None, N, 0, ptr(o32), T, 1, T, -1, 0)
    assert rel_err(o32, ref.reshape(Bn, T, N).permute(0, 2, 1)) < 1e-5


@pytest.mark.parametrize("cfg,col0,N,T,vt_lds", [(8, 128, 192, 64, 1), (9, 256, 384, 256, 1), (0, 256, 384, 64, 1), (1, 128, 192, 256, 1), (5, 512, 768, 1024, 1), (4, 640, 960, 64, 1),
                                                  (0, 256, 384, 64, 0), (4, 512, 768, 256, 1), (16, 512, 768, 1024, 1), (16, 256, 512, 64, 1), (17, 640, 960, 64, 1),
                                                  (17, 640, 960, 4096, 1)])
def test_gemm_transposed_columns_through_lds_epilogue(ctx, cfg, col0, N, T, vt_lds):
    """Fused q|k|v projection: columns >= col0 leave TRANSPOSED per batch item (V^T for the attention kernel).  With tile-aligned
    col0 the LDS epilogue stages those tiles transposed and stores 8-token runs (tokens per item below, equal to and above the tile
    height); (4, 512, ...) is the misaligned case that falls back to the scalar epilogue; vt_lds = 0 forces that fallback."""
    Bn, K = 3, 320
    M = Bn * T
    a, w = h16(M, K, seed=61), h16(N, K, scale=1.0 / math.sqrt(K), seed=62)
    bias = torch.randn(N, device=DEV)
    out = torch.zeros(M, col0, dtype=torch.half, device=DEV)
    vt = torch.zeros(Bn, N - col0, T, dtype=torch.half, device=DEV)
    assert ctx.lib.pnpi_set_tuning(b"igemm_vt_lds", vt_lds) == 0
    try:
        ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, 1.0, ptr(bias), None, ptr(out), col0, col0, ptr(vt), T, 0, T, cfg, 0)
        torch.cuda.synchronize()
    finally:
        ctx.lib.pnpi_set_tuning(b"igemm_vt_lds", 1)
    ref = a.float() @ w.float().t() + bias
    assert rel_err(out, ref[:, :col0]) < 2e-3
    assert rel_err(vt, ref[:, col0:].reshape(Bn, T, N - col0).permute(0, 2, 1)) < 2e-3


def _pp_random_gemm_cases(n, seed):
    """(M, N, K, cfg, split, has_bias, has_res, alpha) drawn from the ranges the ping-pong kernel serves: M to beyond 2^16 (a linear layer's x
    coordinate), N any multiple of 8 (ragged n-tiles, the weight-row clamp), 1 .. 40 K-tiles, split-K with empty last slices."""
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        cfg = rng.choice((16, 17))
        M = rng.choice((rng.randint(1, 600), rng.randint(600, 5000), rng.randint(30000, 70000)))
        N = 8 * rng.randint(1, 170)
        K = 64 * rng.randint(1, 40 if M < 6000 else 6)
        split = rng.choice((0, 0, 0, 2, 3, 5, 7)) if K >= 256 else 0
        out.append((M, N, K, cfg, split, rng.random() < 0.7, rng.random() < 0.5, rng.choice((1.0, 1.0, 0.5))))
    return out


@pytest.mark.parametrize("M,N,K,cfg,split,has_bias,has_res,alpha", _pp_random_gemm_cases(28, 20260926))
def test_pingpong_gemm_random_shapes(ctx, M, N, K, cfg, split, has_bias, has_res, alpha):
    a = h16(M, K, seed=M % 97)
    w = h16(N, K, scale=1.0 / math.sqrt(K), seed=N % 89)
    bias = torch.randn(N, device=DEV) if has_bias else None
    res = h16(M, N, seed=3) if has_res else None
    out = torch.full((M, N), float("nan"), dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_gemm", ptr(a), K, ptr(w), K, M, N, K, alpha, ptr(bias), ptr(res), ptr(out), N, 1 << 30, None, 0, 0, 1, cfg, split)
    ref = alpha * (a.float() @ w.float().t())
    if has_bias:
        ref = ref + bias
    if has_res:
        ref = ref + res.float()
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 2e-3, (rel_err(out, ref), max_err(out, ref))


def _pp_random_conv_cases(n, seed):
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        cfg = rng.choice((16, 17))
        C1 = 64 * rng.randint(1, 4)
        C2 = 64 * rng.randint(0, 2)
        H = rng.choice((7, 8, 12, 16, 24, 33))
        stride, pad, ups = rng.choice(((1, 1, 0), (1, 1, 0), (2, 1, 0), (2, 0, 0), (1, 1, 1)))
        if ups:
            H = min(H, 12)
        B = rng.randint(1, 5)
        N = 8 * rng.randint(4, 90)
        split = rng.choice((0, 0, 2, 4))
        out.append((B, C1, C2, H, N, stride, pad, ups, cfg, split))
    return out


@pytest.mark.parametrize("B,C1,C2,H,N,stride,pad,ups,cfg,split", _pp_random_conv_cases(20, 4242))
def test_pingpong_conv_random_shapes(ctx, B, C1, C2, H, N, stride, pad, ups, cfg, split):
    test_conv3x3(ctx, B, C1, C2, H, N, stride, pad, ups, cfg, split)


@pytest.mark.parametrize("cfg", [16, 17])
@pytest.mark.parametrize("stride,pad,ups,H", [(1, 1, 0, 9), (2, 1, 0, 15), (2, 0, 0, 16), (1, 1, 1, 7)])
def test_pingpong_conv_out_of_range_lanes_write_zeros_into_poisoned_lds(ctx, cfg, stride, pad, ups, H):
    """conv padding, rows past M and tail K-tiles of the ping-pong kernel rely on `buffer_load ... lds` writing ZEROS for lanes
    whose offset is >= num_records.  If such a lane skipped its LDS write, whatever the previous launch left in LDS would be multiplied
    in: so a GEMM with +-30 000 operands runs first on the same stream (every CU's LDS full of them), then a convolution with a padded
    border, M % BM != 0 (B H W = 3 x odd^2), stride 2 / asymmetric pad / folded upsample -- border rows, tail rows and the whole output
    against F.conv2d.  Stale values of that size would turn the result into garbage, not into a rounding difference."""
    big = torch.full((1024, 512), 30000.0, dtype=torch.half, device=DEV)
    big[::2] = -30000.0
    wbig = torch.full((512, 512), 1.0, dtype=torch.half, device=DEV)
    sink = torch.empty(1024, 512, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_gemm", ptr(big), 512, ptr(wbig), 512, 1024, 512, 512, 1e-6, None, None, ptr(sink), 512, 1 << 30, None, 0, 0, 1, cfg, 0)
    test_conv3x3(ctx, 3, 128, 64, H, 328, stride, pad, ups, cfg, 0)
    # and with split-K: the last K slice of a tile is a partial walk
    ctx.call("pnpi_op_gemm", ptr(big), 512, ptr(wbig), 512, 1024, 512, 512, 1e-6, None, None, ptr(sink), 512, 1 << 30, None, 0, 0, 1, cfg, 0)
    test_conv3x3(ctx, 3, 128, 64, H, 328, stride, pad, ups, cfg, 3)


# ------------------------------------------------------------------------------------------------ conv
def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def pack_w(w):  # [N, C, kh, kw] -> [N, kh*kw*C] tap-major
    return w.permute(0, 2, 3, 1).contiguous().reshape(w.shape[0], -1)


@pytest.mark.parametrize("B,C1,C2,H,N,stride,pad,ups,cfg,split", [
    (2, 64, 0, 16, 64, 1, 1, 0, -1, 0),       # fast-K path
    (2, 64, 64, 16, 128, 1, 1, 0, 0, 0),      # concat, 128x128 tile
    (1, 320, 0, 32, 320, 1, 1, 0, -1, 0),     # SD shape, FASTK
    (2, 32, 0, 16, 64, 1, 1, 0, -1, 0),       # Cin not a multiple of 64 -> generic K path
    (2, 8, 0, 16, 32, 1, 1, 0, -1, 0),        # conv_in style (4 channels padded to 8)
    (2, 64, 0, 16, 64, 2, 1, 0, -1, 0),       # UNet downsample (symmetric pad)
    (2, 64, 0, 16, 64, 2, 0, 0, -1, 0),       # VAE downsample (pad (0,1,0,1))
    (2, 64, 0, 8, 64, 1, 1, 1, -1, 0),        # nearest-2x upsample folded into the conv
    (4, 1280, 0, 8, 128, 1, 1, 0, 2, 4),      # deep K, split-K
    (2, 128, 64, 8, 4, 1, 1, 0, -1, 0),       # tiny N (conv_out style), concat with generic... C1=128 fast
    (3, 64, 64, 16, 192, 1, 1, 0, 3, 0),      # 256x128 tile, concat, ragged M (768 rows) and N
    (3, 64, 64, 16, 320, 1, 1, 0, 4, 0),      # 128x320 tile, concat
    (2, 320, 0, 16, 640, 1, 1, 0, 4, 0),      # 128x320 tile, two n-tiles
    (2, 64, 0, 8, 320, 1, 1, 1, 4, 0),        # 128x320 tile with the folded upsample
    (2, 128, 0, 16, 256, 2, 1, 0, 5, 0),      # 128x256 tile, stride 2
    (4, 1280, 0, 8, 320, 1, 1, 0, 4, 6),      # 128x320 tile, deep K, split-K
    (1, 320, 0, 8, 320, 1, 1, 0, 8, 0),       # 64x64 tile, 4 k-groups of waves (K = 2880: 45 chunks over 4 groups, ragged)
    (2, 1280, 0, 8, 128, 1, 1, 0, 9, 0),      # 128x128 tile, 2 k-groups
    (2, 64, 64, 8, 64, 1, 1, 0, 11, 0),       # 64x64 tile, 2 k-groups, concat (tap / source changes inside a group's slice)
    (2, 64, 0, 8, 64, 1, 1, 1, 8, 0),         # 4 k-groups with the folded upsample
    (3, 64, 64, 16, 320, 1, 1, 0, 6, 0),      # 256x320 tile (8 waves), concat, ragged M
    (2, 128, 0, 16, 256, 1, 1, 0, 7, 0),      # 256x256 tile (8 waves)
    (2, 64, 0, 8, 64, 1, 1, 1, 3, 0),         # 256x128 tile with the folded upsample
    (3, 64, 64, 16, 320, 1, 1, 0, 17, 0),     # ping-pong 192x320, concat (source switch inside the k-walk), ragged M
    (1, 320, 0, 32, 320, 1, 1, 0, 17, 0),     # ping-pong 192x320, SD shape (45 K-tiles)
    (2, 128, 0, 16, 256, 1, 1, 0, 16, 0),     # ping-pong 256x256
    (2, 64, 0, 8, 64, 1, 1, 1, 16, 0),        # ping-pong 256x256 with the folded upsample
    (2, 128, 0, 16, 256, 2, 1, 0, 16, 0),     # ping-pong 256x256, stride 2
    (2, 64, 0, 16, 64, 2, 0, 0, 17, 0),       # ping-pong 192x320, VAE downsample (pad (0,1,0,1))
    (4, 1280, 0, 8, 320, 1, 1, 0, 17, 6),     # ping-pong 192x320, deep K, split-K
    (3, 128, 64, 16, 640, 1, 1, 0, 17, 2),    # ping-pong 192x320, concat + split-K (a split starts inside the second source)
])
def test_conv3x3(ctx, B, C1, C2, H, N, stride, pad, ups, cfg, split):
    W = H
    x1 = h16(B, C1, H, W, seed=6)
    x2 = h16(B, C2, H, W, seed=7) if C2 else None
    Cin = C1 + C2
    w = h16(N, Cin, 3, 3, scale=1.0 / math.sqrt(9 * Cin), seed=8)
    bias = torch.randn(N, device=DEV)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    xr = xin.float()
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    if stride == 2 and pad == 0:
        xr = F.pad(xr, (0, 1, 0, 1))
        ref = F.conv2d(xr, w.float(), bias, stride=2, padding=0)
    else:
        ref = F.conv2d(xr, w.float(), bias, stride=stride, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = h16(B, N, Ho, Wo, seed=9)
    ref = ref + res.float()
    out = torch.zeros(B, Ho, Wo, N, dtype=torch.half, device=DEV)
    # keep every device tensor alive in a local until the result has been read back
    x1n, x2n, wp, resn = nhwc(x1), (nhwc(x2) if C2 else None), pack_w(w), nhwc(res)
    ctx.call("pnpi_op_conv", ptr(x1n), ptr(x2n), C1, C2, B, H, W, 3, stride, pad, ups, Ho, Wo,
             ptr(wp), ptr(bias), ptr(resn), N, ptr(out), cfg, split)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-3, (rel_err(got, ref), max_err(got, ref))


@pytest.mark.parametrize("B,C1,C2,H,N,stride,pad,ups,cfg,split", [
    (3, 64, 64, 16, 320, 1, 1, 0, 17, 0), (1, 320, 0, 32, 320, 1, 1, 0, 17, 0), (2, 128, 0, 16, 256, 2, 1, 0, 16, 0), (2, 64, 0, 8, 64, 1, 1, 1, 16, 0),
    (4, 1280, 0, 8, 320, 1, 1, 0, 17, 6), (3, 128, 64, 16, 640, 1, 1, 0, 17, 2)])
def test_conv3x3_channel_slab_major_walk(ctx, B, C1, C2, H, N, stride, pad, ups, cfg, split):
    """The ping-pong kernel's other k order (tuning igemm_tapin: the nine filter taps of a 64-channel slab in a row, GemmP::k_order): same
    products, another summation order -- concat sources, stride 2, the folded upsample and split-K ranges of the permuted walk."""
    assert ctx.lib.pnpi_set_tuning(b"igemm_tapin", 1) == 0
    try:
        test_conv3x3(ctx, B, C1, C2, H, N, stride, pad, ups, cfg, split)
    finally:
        ctx.lib.pnpi_set_tuning(b"igemm_tapin", 0)


def test_conv1x1_concat(ctx):
    B, C1, C2, H, N = 2, 128, 64, 8, 96
    x1, x2 = h16(B, C1, H, H, seed=10), h16(B, C2, H, H, seed=11)
    w = h16(N, C1 + C2, 1, 1, scale=0.08, seed=12)
    bias = torch.randn(N, device=DEV)
    ref = F.conv2d(torch.cat([x1, x2], 1).float(), w.float(), bias)
    out = torch.zeros(B, H, H, N, dtype=torch.half, device=DEV)
    x1n, x2n, wp = nhwc(x1), nhwc(x2), pack_w(w)
    ctx.call("pnpi_op_conv", ptr(x1n), ptr(x2n), C1, C2, B, H, H, 1, 1, 0, 0, H, H, ptr(wp), ptr(bias), None, N,
             ptr(out), -1, 0)
    torch.cuda.synchronize()
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 2e-3


# ------------------------------------------------------------------------------------------------ norms / elementwise
@pytest.mark.parametrize("B,C1,C2,HW,silu,eps", [(2, 320, 0, 256, 1, 1e-5), (2, 64, 32, 64, 1, 1e-6), (3, 32, 0, 16, 0, 1e-6),
                                                 (1, 1280, 1280, 64, 1, 1e-5), (1, 128, 0, 4096, 1, 1e-6),
                                                 # the one-launch kernel (HW * C / 32 <= 24576): 5 of its 6 register slices, a group that
                                                 # straddles the two concat sources, the largest slice, a 2 x 2 map
                                                 (2, 640, 0, 1024, 1, 1e-5), (1, 1280, 640, 256, 1, 1e-5), (3, 768, 0, 1024, 0, 1e-6),
                                                 (2, 1280, 0, 4, 1, 1e-5), (12, 2560, 0, 64, 1, 1e-5)])
def test_groupnorm(ctx, B, C1, C2, HW, silu, eps):
    C = C1 + C2
    x1 = (h16(B, HW, C1, seed=13).float() * 2 + 0.5).half()
    x2 = h16(B, HW, C2, seed=14) if C2 else None
    gamma = torch.randn(C, device=DEV) * 0.2 + 1
    beta = torch.randn(C, device=DEV) * 0.2
    out = torch.zeros(B, HW, C, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_groupnorm", ptr(x1), ptr(x2) if C2 else None, C1, C2, B, HW, 32, eps, ptr(gamma), ptr(beta), silu, ptr(out))
    xin = x1 if x2 is None else torch.cat([x1, x2], 2)
    ref = F.group_norm(xin.float().permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    assert rel_err(out.permute(0, 2, 1), ref) < 2e-3


@pytest.mark.parametrize("M,C", [(1024, 320), (100, 1280), (7, 32), (64, 640)])
def test_layernorm(ctx, M, C):
    x = (h16(M, C, seed=15).float() * 3 + 1).half()
    gamma = torch.randn(C, device=DEV) * 0.2 + 1
    beta = torch.randn(C, device=DEV) * 0.2
    out = torch.zeros(M, C, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_layernorm", ptr(x), M, C, 1e-5, ptr(gamma), ptr(beta), ptr(out))
    assert rel_err(out, F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)) < 2e-3


def ilv32(a, g):
    """[.., I] x and gate -> [.., 2I] packed as [x(32) | gate(32)] groups (the layout of the fused GEGLU epilogue)."""
    I = a.shape[-1]
    return torch.stack([a.reshape(*a.shape[:-1], I // 32, 32), g.reshape(*g.shape[:-1], I // 32, 32)], dim=-2).reshape(*a.shape[:-1], 2 * I)


def test_geglu(ctx):
    M, I = 333, 256
    x = h16(M, 2 * I, scale=2.0, seed=16)
    a, g = x.float().chunk(2, dim=-1)
    xi = ilv32(a, g).half().contiguous()
    out = torch.zeros(M, I, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_geglu", ptr(xi), M, I, ptr(out))
    assert rel_err(out, a * F.gelu(g)) < 2e-3


@pytest.mark.parametrize("force", [-1, 0, 1, 4, 5, 6, 8, 9, 12, 13, 14, 15, 16, 17])
@pytest.mark.parametrize("M,K,I", [(4096, 320, 1280), (1024, 640, 2560), (200, 1280, 5120), (64, 64, 64)])
def test_gemm_fused_geglu(ctx, M, K, I, force):
    """GEGLU.forward (my_diffusers/models/attention.py:331-333): proj -> chunk -> x * gelu(gate), fused into the GEMM epilogue
    of every tile configuration (force = tile configuration of launch_igemm; -1 = the cost model's choice)."""
    assert ctx.lib.pnpi_set_tuning(b"igemm_force_cfg", force) == 0
    try:
        _geglu_case(ctx, M, K, I)
    finally:
        ctx.lib.pnpi_set_tuning(b"igemm_force_cfg", -1)


def _geglu_case(ctx, M, K, I):
    a = h16(M, K, seed=51)
    w = h16(2 * I, K, scale=1.0 / math.sqrt(K), seed=52)
    bias = torch.randn(2 * I, device=DEV)
    wx, wg = w[:I], w[I:]
    wi = ilv32(wx.t(), wg.t()).t().contiguous()            # interleave rows
    bi = ilv32(bias[:I][None], bias[I:][None])[0].contiguous()
    out = torch.zeros(M, I, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_gemm_geglu", ptr(a), K, ptr(wi), K, M, 2 * I, K, ptr(bi), ptr(out), I)
    h = a.float() @ w.float().t() + bias
    ref = h[:, :I] * F.gelu(h[:, I:])
    # the projection is rounded to fp16 once before the product (as the unfused reference path in fp16 would)
    assert rel_err(out, ref) < 3e-3, rel_err(out, ref)


def test_softmax_rows(ctx):
    M, N = 37, 4096
    x = h16(M, N, scale=3.0, seed=17)
    ref = x.float().softmax(-1)
    ctx.call("pnpi_op_softmax_rows", ptr(x), M, N, N)
    assert rel_err(x, ref) < 2e-3


# ------------------------------------------------------------------------------------------------ attention
def make_qkv(B, heads, Nq, Nk, dh, Dp, seed):
    q = h16(B, heads, Nq, dh, seed=seed)
    k = h16(B, heads, Nk, dh, seed=seed + 1)
    v = h16(B, heads, Nk, dh, seed=seed + 2)
    hd = heads * Dp
    qb = torch.zeros(B, Nq, hd, dtype=torch.half, device=DEV)
    kb = torch.zeros(B, Nk, hd, dtype=torch.half, device=DEV)
    ldv = (Nk + 7) // 8 * 8
    vt = torch.full((B, heads, Dp, ldv), float("nan"), dtype=torch.half, device=DEV)  # pads are NaN on purpose
    vt[:, :, :, :Nk] = 0
    for h in range(heads):
        qb[:, :, h * Dp:h * Dp + dh] = q[:, h]
        kb[:, :, h * Dp:h * Dp + dh] = k[:, h]
        vt[:, h, :dh, :Nk] = v[:, h].transpose(1, 2)
    return q, k, v, qb, kb, vt, ldv


@pytest.mark.parametrize("B,heads,Nq,Nk,dh,Dp", [
    (2, 2, 256, 256, 40, 64), (1, 2, 1024, 1024, 80, 96), (2, 8, 64, 64, 160, 160), (2, 2, 200, 77, 40, 64),
    (1, 2, 16, 16, 8, 32), (1, 1, 100, 130, 32, 32), (1, 2, 4096, 4096, 40, 64), (2, 2, 64, 77, 160, 160), (1, 2, 96, 77, 128, 128),
])
def test_attention_flash(ctx, B, heads, Nq, Nk, dh, Dp):
    q, k, v, qb, kb, vt, ldv = make_qkv(B, heads, Nq, Nk, dh, Dp, seed=20)
    scale = dh ** -0.5
    o = torch.zeros(B, Nq, heads * dh, dtype=torch.half, device=DEV)
    rows = torch.arange(B, dtype=torch.int32, device=DEV).repeat_interleave(4).reshape(B, 4).contiguous()
    ctx.call("pnpi_op_attention", ptr(qb), heads * Dp, 0, ptr(kb), heads * Dp, 0, ptr(vt), ldv, ptr(o), heads * dh, heads, Nq, Nk,
             Dp, dh, scale, ptr(rows), B)
    ref = (q.float() @ k.float().transpose(-1, -2) * scale).softmax(-1) @ v.float()   # [B, heads, Nq, dh]
    ref = ref.permute(0, 2, 1, 3).reshape(B, Nq, heads * dh)
    assert torch.isfinite(o.float()).all()
    assert rel_err(o, ref) < 3e-3, (rel_err(o, ref), max_err(o, ref))


@pytest.mark.parametrize("B,heads,N", [(2, 2, 256), (1, 8, 4096)])
def test_attention_flash_permuted_vt_and_its_producer(ctx, B, heads, N):
    """The 4096-token sites: V^T in the permuted key order (each aligned 16-token group stored [0-3, 8-11, 4-7, 12-15], ops.h
    vt_perm16_pos) read by the 64-wide LDS-DMA kernel with one 16-byte fragment load; and the producer -- a projection GEMM whose
    transposed epilogue writes that order -- against a plain transpose."""
    dh, Dp = 40, 64
    q, k, v, qb, kb, vt, ldv = make_qkv(B, heads, N, N, dh, Dp, seed=25)
    pos = torch.arange(N, device=DEV)
    swap = (((pos >> 2) ^ (pos >> 3)) & 1).bool()
    perm_pos = torch.where(swap, pos ^ 12, pos)                 # token t lives at position perm_pos[t]
    vtp = torch.empty_like(vt)
    vtp[..., perm_pos] = vt[..., pos]
    scale = dh ** -0.5
    o = torch.zeros(B, N, heads * dh, dtype=torch.half, device=DEV)
    rows = torch.arange(B, dtype=torch.int32, device=DEV).repeat_interleave(4).reshape(B, 4).contiguous()
    assert ctx.lib.pnpi_set_tuning(b"op_attention_vt_perm", 1) == 0
    try:
        ctx.call("pnpi_op_attention", ptr(qb), heads * Dp, 0, ptr(kb), heads * Dp, 0, ptr(vtp), ldv, ptr(o), heads * dh, heads, N, N, Dp, dh,
                 scale, ptr(rows), B)
    finally:
        ctx.lib.pnpi_set_tuning(b"op_attention_vt_perm", 0)
    ref = ((q.float() @ k.float().transpose(-1, -2) * scale).softmax(-1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, heads * dh)
    assert rel_err(o, ref) < 3e-3, rel_err(o, ref)
    o2 = torch.zeros_like(o)
    ctx.call("pnpi_op_attention", ptr(qb), heads * Dp, 0, ptr(kb), heads * Dp, 0, ptr(vt), ldv, ptr(o2), heads * dh, heads, N, N, Dp, dh, scale,
             ptr(rows), B)
    assert torch.equal(o, o2)          # the same numbers in the same order: bit-identical to the plain layout


@pytest.mark.parametrize("B,heads,N,perm", [(1, 8, 4096, 1), (2, 2, 256, 0), (1, 2, 1024, 1)])
def test_attention_flash_augmented_column(ctx, B, heads, N, perm):
    """The 64-wide LDS-DMA kernel with AttnP::aug (d = 40 heads): column 40 of every K row and row 40 of V^T hold 1.0 (in the model the q | k | v
    projection's bias writes them); the kernel then takes the running-max shift through Q's column 40 and the row sum through V^T's row 40
    (no per-element fma / add).  Against fp32 softmax attention, against the plain kernel on the same inputs, and with keys that force the
    reference point to move late in the key walk (a spike at key 3 * N / 4) and queries whose scores are all far below zero (first-tile path)."""
    dh, Dp = 40, 64
    q, k, v, qb, kb, vt, ldv = make_qkv(B, heads, N, N, dh, Dp, seed=27)
    # a late spike: key j0 is 6 x query i0 -> its logit is far above everything seen before it; and a query far from every key
    i0, j0 = 5, (3 * N) // 4
    k[:, :, j0] = 6.0 * q[:, :, i0]
    q[:, :, 9] = -8.0 * k[:, :, :64].mean(dim=2)
    kb.view(B, N, heads, Dp)[..., :dh] = k.permute(0, 2, 1, 3)
    qb.view(B, N, heads, Dp)[..., :dh] = q.permute(0, 2, 1, 3)
    scale = dh ** -0.5
    ref = ((q.float() @ k.float().transpose(-1, -2) * scale).softmax(-1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, heads * dh)
    rows = torch.arange(B, dtype=torch.int32, device=DEV).repeat_interleave(4).reshape(B, 4).contiguous()
    pos = torch.arange(N, device=DEV)
    vtx = vt.clone()
    if perm:
        swap = (((pos >> 2) ^ (pos >> 3)) & 1).bool()
        vtx[..., torch.where(swap, pos ^ 12, pos)] = vt[..., pos]

    def run(aug):
        kk, vv = kb.clone(), vtx.clone()
        if aug:
            kk.view(B, N, heads, Dp)[..., dh] = 1.0
            vv.view(B, heads, Dp, -1)[:, :, dh, :] = 1.0
        o = torch.zeros(B, N, heads * dh, dtype=torch.half, device=DEV)
        assert ctx.lib.pnpi_set_tuning(b"op_attention_aug", aug) == 0 and ctx.lib.pnpi_set_tuning(b"op_attention_vt_perm", perm) == 0
        try:
            ctx.call("pnpi_op_attention", ptr(qb), heads * Dp, 0, ptr(kk), heads * Dp, 0, ptr(vv), ldv, ptr(o), heads * dh, heads, N, N, Dp, dh, scale,
                     ptr(rows), B)
            torch.cuda.synchronize()
        finally:
            ctx.lib.pnpi_set_tuning(b"op_attention_aug", 0)
            ctx.lib.pnpi_set_tuning(b"op_attention_vt_perm", 0)
        return o

    plain, aug = run(0), run(1)
    assert torch.isfinite(aug).all()
    assert rel_err(plain, ref) < 3e-3, rel_err(plain, ref)
    assert rel_err(aug, ref) < 3e-3, rel_err(aug, ref)
    assert rel_err(aug, plain.float()) < 2e-3, rel_err(aug, plain.float())
    # the spiked query row and the far-away query row individually
    for i in (i0, 9):
        assert rel_err(aug[:, i], ref[:, i]) < 5e-3, (i, rel_err(aug[:, i], ref[:, i]))


def test_attention_row_indirection(ctx):
    # self-attention replacement: output row 3 uses q,k of row 2 and its own v (attention_control.py:258-263)
    B, heads, N, dh, Dp = 4, 2, 128, 40, 64
    q, k, v, qb, kb, vt, ldv = make_qkv(B, heads, N, N, dh, Dp, seed=30)
    scale = dh ** -0.5
    rows = torch.tensor([[0, 0, 0, 0], [1, 1, 1, 1], [2, 2, 2, 2], [3, 2, 2, 3]], dtype=torch.int32, device=DEV)
    o = torch.zeros(B, N, heads * dh, dtype=torch.half, device=DEV)
    ctx.call("pnpi_op_attention", ptr(qb), heads * Dp, 0, ptr(kb), heads * Dp, 0, ptr(vt), ldv, ptr(o), heads * dh, heads, N, N, Dp,
             dh, scale, ptr(rows), B)
    p = (q.float() @ k.float().transpose(-1, -2) * scale).softmax(-1)
    p[3] = p[2]
    ref = (p @ v.float()).permute(0, 2, 1, 3).reshape(B, N, heads * dh)
    assert rel_err(o, ref) < 3e-3


@pytest.mark.parametrize("Nq,dh,Dp,with_lb", [(256, 160, 160, True), (1024, 80, 96, False), (4096, 40, 64, False), (64, 8, 32, True)])
def test_cross_edit(ctx, Nq, dh, Dp, with_lb):
    # rows: [unc_src, unc_tgt, cond_src, cond_tgt]; the kernel handles the pair (2, 3)
    B, heads, Nk = 4, 2, 77
    q, k, v, qb, kb, vt, ldv = make_qkv(B, heads, Nq, Nk, dh, Dp, seed=40)
    scale = dh ** -0.5
    g = torch.Generator().manual_seed(5)
    mm = torch.eye(77)
    mm[3, 3] = 0; mm[3, 4] = 0.5; mm[3, 5] = 0.5; mm[10] = 0; mm[10, 76] = 1.0    # some non-trivial mapping
    c1 = torch.rand(96, generator=g); c2 = torch.rand(96, generator=g)
    lba = torch.zeros(2, 96); lba[0, 2] = 1; lba[1, 2] = 1; lba[1, 3] = 1
    mmT = torch.zeros(96, 96); mmT[:77, :77] = mm.t()
    mmT16 = mmT.half().to(DEV).contiguous()
    pairs = torch.tensor([[2, 3]], dtype=torch.int32, device=DEV)
    nslots = 2 * heads
    lb_acc = torch.ones(1, nslots, 2, Nq, device=DEV)
    o = torch.zeros(B, Nq, heads * dh, dtype=torch.half, device=DEV)
    c1d, c2d, lbad = c1.to(DEV), c2.to(DEV), lba.to(DEV).contiguous()
    ctx.call("pnpi_op_cross_edit", ptr(qb), heads * Dp, 0, ptr(kb), heads * Dp, 0, ptr(vt), ldv, ptr(o), heads * dh, heads, Nq, Nk, Dp,
             dh, scale, ptr(pairs), 1, ptr(mmT16), ptr(c1d), ptr(c2d), ptr(lbad) if with_lb else None,
             ptr(lb_acc) if with_lb else None, heads, nslots)
    torch.cuda.synchronize()
    p = (q.float() @ k.float().transpose(-1, -2) * scale).softmax(-1)        # [B, heads, Nq, 77]
    psrc, ptgt = p[2], p[3]
    mapped = psrc @ mm.to(DEV)
    pnew = c1[:77].to(DEV) * mapped + c2[:77].to(DEV) * ptgt
    ref_src = (psrc @ v[2].float()).permute(1, 0, 2).reshape(Nq, heads * dh)
    ref_tgt = (pnew @ v[3].float()).permute(1, 0, 2).reshape(Nq, heads * dh)
    assert rel_err(o[2], ref_src) < 3e-3, rel_err(o[2], ref_src)
    assert rel_err(o[3], ref_tgt) < 4e-3, rel_err(o[3], ref_tgt)
    assert o[0].abs().max() == 0 and o[1].abs().max() == 0        # rows outside the pair are not touched
    if with_lb:
        la = lba[:, :77].to(DEV)
        exp_src = 1 + (psrc * la[0]).sum(-1)      # [heads, Nq]
        exp_tgt = 1 + (pnew * la[1]).sum(-1)
        got = lb_acc[0, heads:2 * heads]           # slots lb_slot0 + head
        assert rel_err(got[:, 0], exp_src) < 3e-3
        assert rel_err(got[:, 1], exp_tgt) < 3e-3
        assert (lb_acc[0, :heads] == 1).all()


# ------------------------------------------------------------------------------------------------ step kernels (bit exact)
def test_step_kernels_bit_exact(ctx):
    """DDIM moves / CFG / the direct-inversion lines are bit-exact against the oracle (oracle/p2p_oracle.py, which evaluates
    the reference formulas with correctly rounded fp32 scalars) and within 1e-6 of the formulas as torch evaluates them."""
    from oracle import p2p_oracle as po
    ac = po.alphas_cumprod()
    final = ac[0]
    arr = (C.c_float * 1000)(*ac.tolist())
    ctx.call("pnpi_set_scheduler", arr, 1000, float(final))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 4, 16, 16, generator=g)
    e4 = torch.randn(4, 4, 16, 16, generator=g)
    xd, e4d = x.to(DEV), e4.to(DEV)
    for t in (980, 500, 480, 20, 0):
        # next_step (inversion.py:262-270)
        e = e4[:2].contiguous()
        a_f, a_n = po.next_alphas(ac, final, t, 20)
        ref = po.ddim_move(x, e, float(a_f), float(a_n))
        out = torch.empty(2, 4, 16, 16, device=DEV)
        ed = e.to(DEV)
        ctx.call("pnpi_ddim_next_step", ptr(ed), t, 20, ptr(xd), x.numel(), ptr(out))
        assert torch.equal(out.cpu(), ref), t
        x0 = (x - (1 - a_f) ** 0.5 * e) / a_f ** 0.5
        assert torch.allclose(out.cpu(), a_n ** 0.5 * x0 + (1 - a_n) ** 0.5 * e, atol=2e-6, rtol=0)
        # CFG + prev_step + offset (inversion.py:383-389)
        a_t, a_p = po.prev_alphas(ac, final, t, 20)
        eu, ec = e4.chunk(2)
        eg = eu + 7.5 * (ec - eu)
        prev = po.ddim_move(x, eg, float(a_t), float(a_p))
        target = torch.randn(1, 4, 16, 16, generator=g)
        loss = target - prev
        cur = prev + loss
        off = torch.empty(2, 4, 16, 16, device=DEV)
        xo = torch.empty(2, 4, 16, 16, device=DEV)
        td = target.to(DEV)
        ctx.call("pnpi_cfg_ddim_prev", ptr(e4d), ptr(xd), 1, 2, x[0].numel(), 7.5, t, 20, None, 0, ptr(td), 1.0, ptr(off), ptr(xo), None, 0, None)
        assert torch.equal(off.cpu(), loss) and torch.equal(xo.cpu(), cur), t
        # guidance step with noise_loss on the first row only (p2p_guidance_forward.py:110-114)
        nl = torch.randn(2, 4, 16, 16, generator=g)
        ref2 = torch.cat((prev[:1] + nl[:1], prev[1:]))
        nld = nl.to(DEV)
        ctx.call("pnpi_cfg_ddim_prev", ptr(e4d), ptr(xd), 1, 2, x[0].numel(), 7.5, t, 20, ptr(nld), 1, None, 1.0, None, ptr(xo), None, 0, None)
        assert torch.equal(xo.cpu(), ref2), t
        # DDIMSchedulerDev.step == prev_step
        ctx.call("pnpi_ddim_prev_step", ptr(ed), t, 20, ptr(xd), x.numel(), ptr(out))
        assert torch.equal(out.cpu(), po.ddim_move(x, e, float(a_t), float(a_p))), t


def test_local_blend(ctx):
    nslots, mhw, lhw, Cc = 6, 16, 64, 4
    g = torch.Generator().manual_seed(3)
    acc = torch.rand(1, nslots, 2, mhw * mhw, generator=g)
    lat = torch.randn(1, 2, Cc, lhw, lhw, generator=g)
    d_lat = lat.clone().to(DEV)
    accd = acc.to(DEV)
    ctx.call("pnpi_op_local_blend", ptr(accd), nslots, mhw, lhw, Cc, 0.3, ptr(d_lat), 1)
    # LocalBlend.get_mask / __call__ (attention_control.py:97-121)
    maps = acc[0].permute(1, 0, 2).reshape(2, nslots, 1, mhw, mhw).mean(1)
    m = F.max_pool2d(maps, (3, 3), (1, 1), padding=(1, 1))
    m = F.interpolate(m, size=(lhw, lhw))
    m = m / m.max(2, keepdims=True)[0].max(3, keepdims=True)[0]
    m = m.gt(0.3)
    m = m[:1] + m
    x_t = lat[0]
    ref = x_t[:1] + m.float() * (x_t - x_t[:1])
    assert torch.equal(d_lat.cpu()[0], ref)


def test_local_blend_substruct_words(ctx):
    """LocalBlend with substruct_words (attention_control.py:97-118): mask = get_mask(alpha_layers, pooled, th[0]) *
    ~get_mask(substruct_layers, NOT pooled, th[1]); planes 0, 1 of the accumulator are the blend maps, planes 2, 3 the substruct maps."""
    nslots, mhw, lhw, Cc = 6, 16, 64, 4
    g = torch.Generator().manual_seed(5)
    acc = torch.rand(1, nslots, 4, mhw * mhw, generator=g)
    acc[:, :, 2:] *= torch.rand(1, 1, 2, mhw * mhw, generator=g) ** 3          # peaked substruct maps: a real fraction below th[1]
    lat = torch.randn(1, 2, Cc, lhw, lhw, generator=g)
    d_lat, accd = lat.clone().to(DEV), acc.to(DEV)
    th, th_sub = 0.3, 0.45
    ctx.call("pnpi_op_local_blend_sub", ptr(accd), nslots, mhw, lhw, Cc, th, th_sub, ptr(d_lat), 1)

    def get_mask(maps, use_pool, t):
        if use_pool:
            maps = F.max_pool2d(maps, (3, 3), (1, 1), padding=(1, 1))
        m = F.interpolate(maps, size=(lhw, lhw))
        m = m / m.max(2, keepdims=True)[0].max(3, keepdims=True)[0]
        m = m.gt(t)
        return m[:1] + m
    planes = acc[0].permute(1, 0, 2).reshape(4, nslots, 1, mhw, mhw).mean(1)
    mask = get_mask(planes[:2], True, th)
    sub = get_mask(planes[2:], False, th_sub)
    assert 0.05 < sub.float().mean() < 0.95 and 0.05 < (mask * ~sub).float().mean() < 0.95
    x_t = lat[0]
    ref = x_t[:1] + (mask * ~sub).float() * (x_t - x_t[:1])
    assert torch.equal(d_lat.cpu()[0], ref)


@pytest.mark.parametrize("prox", ["l0", "l1"])
def test_proximal_step_bit_exact(ctx, prox):
    """pnpi_prox_threshold = torch.quantile(|eps_c - eps_u|, q) (sort + linear interpolation) and the soft-threshold inside the
    CFG / DDIM-step kernel (proximal_guidance_forward.py:39-64), bit for bit against the oracle formulas on the same eps."""
    from oracle import p2p_oracle as po
    g = torch.Generator().manual_seed(11)
    eps = torch.randn(4, 4, 64, 64, generator=g)                    # [unc_src, unc_tgt, cond_src, cond_tgt]
    x = torch.randn(2, 4, 64, 64, generator=g)
    ac, t, ratio = po.alphas_cumprod(), 481, 20
    ctx.call("pnpi_set_scheduler", (C.c_float * 1000)(*ac.tolist()), 1000, float(ac[0]))
    ed, xd = eps.cuda(), x.cuda()
    thr_d = torch.empty(1, device="cuda")
    xo = torch.empty_like(xd)
    for q in (0.75, 0.7, 0.31):
        ctx.call("pnpi_prox_threshold", ptr(ed), 1, 2, x[0].numel(), q, ptr(thr_d))
        d = eps[2:] - eps[:2]
        thr = d.abs().quantile(q)
        assert torch.equal(thr_d.cpu()[0], thr), (q, thr_d.item(), thr.item())
        ctx.call("pnpi_cfg_ddim_prev", ptr(ed), ptr(xd), 1, 2, x[0].numel(), 7.5, t, ratio, None, 0, None, 1.0, None, ptr(xo),
                 ptr(thr_d), 1 if prox == "l0" else 2, None)
        sd = d - d.clamp(-thr, thr)
        if prox == "l1":
            sd = torch.where(sd > 0, sd - thr, sd)
            sd = torch.where(sd < 0, sd + thr, sd)
        e = eps[:2] + 7.5 * sd
        a_t, a_p = po.prev_alphas(ac, ac[0], t, ratio)
        want = po.ddim_move(x, e, float(a_t), float(a_p))
        assert torch.equal(xo.cpu(), want), (xo.cpu() - want).abs().max()


@pytest.mark.parametrize("lr", [0.1, -0.25])
@pytest.mark.parametrize("with_ref", [False, True])
def test_inversion_guidance_step_bit_exact(ctx, with_ref, lr):
    """Level 1 of the inversion pull: pnpi_cfg_ddim_prev with pnpi_recon_desc::inv_x_stars pointing at THIS step's x*_{t-1}
    -- prev - recon_lr * (prev - x*) * (1 - mask_edit) (proximal_guidance_forward.py:73-75) on top of the optional pred-x0 pull, bit for bit
    against the torch formulas; the pred-x0 pull needs recon_lr > 0 (scheduler_dev.py:68), the inversion pull does not; inert outside the
    recon_t window; a descriptor whose struct_size is not this library's is refused."""
    from oracle import p2p_oracle as po
    from pnpinversion_amd import _capi
    g = torch.Generator().manual_seed(21)
    S = ctx.cfg.sample_size
    eps = torch.randn(4, 4, S, S, generator=g)
    x = torch.randn(2, 4, S, S, generator=g)
    ref = torch.randn(1, 4, S, S, generator=g)
    xstar = torch.randn(1, 4, S, S, generator=g)
    ac, ratio, dil = po.alphas_cumprod(), 20, 1
    ctx.call("pnpi_set_scheduler", (C.c_float * 1000)(*ac.tolist()), 1000, float(ac[0]))
    ed, xd, rd, sd_ = eps.cuda(), x.cuda(), ref.cuda(), xstar.cuda()
    thr_d = torch.empty(1, device="cuda")
    ctx.call("pnpi_prox_threshold", ptr(ed), 1, 2, x[0].numel(), 0.75, ptr(thr_d))
    thr = thr_d.cpu()[0]
    d = eps[2:] - eps[:2]
    sd = d - d.clamp(-thr, thr)
    e = eps[:2] + 7.5 * sd
    recon_mask = 1 - F.max_pool2d((sd.abs() > thr).float(), 2 * dil + 1, 1, dil)
    desc = _capi.ReconDesc.make(rd.data_ptr() if with_ref else None, lr, 400, dil, sd_.data_ptr())
    for t, active in ((381, True), (401, False)):
        xo = torch.empty_like(xd)
        ctx.call("pnpi_cfg_ddim_prev", ptr(ed), ptr(xd), 1, 2, x[0].numel(), 7.5, t, ratio, None, 0, None, 1.0, None, ptr(xo),
                 ptr(thr_d), 1, C.byref(desc))
        a_t, a_p = po.prev_alphas(ac, ac[0], t, ratio)
        sa_f, sb_f, sa_t, sb_t = po._scalars(float(a_t), float(a_p), torch.float32)
        x0 = (x - sb_f * e) / sa_f
        if active and with_ref and lr > 0:
            x0 = x0 - lr * (x0 - ref.expand_as(x0)) * recon_mask
        want = sa_t * x0 + sb_t * e
        if active:
            want = want - lr * (want - xstar.expand_as(want)) * recon_mask          # both rows of the image towards the one x*_{t-1}
        assert torch.equal(xo.cpu(), want), (t, (xo.cpu() - want).abs().max())
    bad = _capi.ReconDesc.make(None, lr, 400, dil, sd_.data_ptr())
    bad.struct_size = 32                                                             # e.g. a caller compiled against the four-field struct
    with pytest.raises(_capi.PnpiError, match="struct_size"):
        ctx.call("pnpi_cfg_ddim_prev", ptr(ed), ptr(xd), 1, 2, x[0].numel(), 7.5, 381, ratio, None, 0, None, 1.0, None, ptr(xo),
                 ptr(thr_d), 1, C.byref(bad))


@pytest.mark.parametrize("dil", [0, 1, 2])
@pytest.mark.parametrize("prox", ["l0", "l1"])
def test_reconstruction_guidance_step_bit_exact(ctx, prox, dil):
    """Reconstruction guidance (proximal_guidance_forward.py:48-51,60-72 + DDIMSchedulerDev.step's ref_image branch,
    scheduler_dev.py:68-76) inside the CFG / DDIM-step kernel: mask_edit = dilate(|shrunk delta| > thr), pred_x0 pulled towards the
    encoded source image where the mask is off -- bit for bit against the reference's formulas; inert outside the recon_t window."""
    from oracle import p2p_oracle as po
    from pnpinversion_amd import _capi
    g = torch.Generator().manual_seed(12)
    S = ctx.cfg.sample_size                                     # the library takes the plane size from its model configuration
    eps = torch.randn(4, 4, S, S, generator=g)
    x = torch.randn(2, 4, S, S, generator=g)
    ref = torch.randn(1, 4, S, S, generator=g)
    ac, ratio, lr = po.alphas_cumprod(), 20, 0.1
    ctx.call("pnpi_set_scheduler", (C.c_float * 1000)(*ac.tolist()), 1000, float(ac[0]))
    ed, xd, rd = eps.cuda(), x.cuda(), ref.cuda()
    thr_d = torch.empty(1, device="cuda")
    ctx.call("pnpi_prox_threshold", ptr(ed), 1, 2, x[0].numel(), 0.75, ptr(thr_d))
    thr = thr_d.cpu()[0]
    d = eps[2:] - eps[:2]
    sd = d - d.clamp(-thr, thr)
    if prox == "l1":
        sd = torch.where(sd > 0, sd - thr, sd)
        sd = torch.where(sd < 0, sd + thr, sd)
    e = eps[:2] + 7.5 * sd
    mask_edit = (sd.abs() > thr).float()
    if dil > 0:
        mask_edit = F.max_pool2d(mask_edit, 2 * dil + 1, 1, dil)
    recon_mask = 1 - mask_edit
    desc = _capi.ReconDesc.make(rd.data_ptr(), lr, 400, dil)
    for t, active in ((381, True), (401, False)):
        xo = torch.empty_like(xd)
        ctx.call("pnpi_cfg_ddim_prev", ptr(ed), ptr(xd), 1, 2, x[0].numel(), 7.5, t, ratio, None, 0, None, 1.0, None, ptr(xo),
                 ptr(thr_d), 1 if prox == "l0" else 2, C.byref(desc))
        a_t, a_p = po.prev_alphas(ac, ac[0], t, ratio)
        sa_f, sb_f, sa_t, sb_t = po._scalars(float(a_t), float(a_p), torch.float32)
        x0 = (x - sb_f * e) / sa_f
        if active:
            x0 = x0 - lr * (x0 - ref.expand_as(x0)) * recon_mask
        want = sa_t * x0 + sb_t * e
        assert torch.equal(xo.cpu(), want), (t, (xo.cpu() - want).abs().max())
    assert 0.0 < recon_mask.mean().item() < 1.0

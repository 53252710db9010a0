"""The 64 x 64-level self-attention launch of the 12-row forward (12 rows x 8 heads x 4096 tokens, d = 40) through pnpi_op_attention with the
augmented column, timed alone (PNPI_ATTN_NOAUG=1: the kernel without the in-MFMA max shift / row sum).  usage: attn_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr  # noqa: E402
ctx = Ctx(); lib = ctx.lib
B, heads, N, dh, Dp = 12, 8, 4096, 40, 64
hd = heads * Dp
qk = torch.zeros(B, N, 2 * hd, device="cuda", dtype=torch.half)
qk.view(B, N, 2, heads, Dp)[..., :dh] = torch.randn(B, N, 2, heads, dh, device="cuda").half()
qk.view(B, N, 2, heads, Dp)[:, :, 1, :, dh] = 1.0
vt = torch.zeros(B, heads, Dp, N, device="cuda", dtype=torch.half)
vt[:, :, :dh] = torch.randn(B, heads, dh, N, device="cuda").half(); vt[:, :, dh] = 1.0
o = torch.zeros(B, N, heads * dh, device="cuda", dtype=torch.half)
rows = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(4).reshape(B, 4).contiguous()
assert lib.pnpi_set_tuning(b"op_attention_aug", 1) == 0 and lib.pnpi_set_tuning(b"op_attention_vt_perm", 1) == 0
f = lambda: ctx.call("pnpi_op_attention", ptr(qk), 2 * hd, 0, ptr(qk), 2 * hd, hd, ptr(vt), N, ptr(o), heads * dh, heads, N, N, Dp, dh, dh ** -0.5, ptr(rows), B)
for _ in range(3): f()
torch.cuda.synchronize()
best = 1e9
for rnd in range(5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 10 * 1e3)
fl = 4.0 * B * heads * N * N * dh
print("PNPI_ATTN_NOAUG=%s: %.1f us per launch, %.0f TFLOP/s algorithmic, checksum %.4f" % (os.environ.get("PNPI_ATTN_NOAUG", "0"), best, fl / best / 1e6, float(o.float().abs().mean())))

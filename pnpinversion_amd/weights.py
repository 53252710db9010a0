"""Synthetic, seeded Stable-Diffusion-1.x weights in the diffusers state-dict layout (keys of UNet2DConditionModel /
AutoencoderKL, identical in diffusers 0.3.0 .. 0.10.0 for this architecture; see models/edict/my_diffusers/models/*).

There are no SD checkpoints on the build / GPU boxes (no network), so parity and benchmarks use these.  Every tensor is drawn
from a counter-based generator keyed by (seed, key name), so any process regenerates the identical value without shipping
files, and every value is exactly representable in fp16 (the device arithmetic type), which keeps the CPU oracle and the HIP
path on bit-identical parameters.  A real checkpoint loads through the same `load_state_dict` path unchanged."""
import zlib

import numpy as np
import torch

from .config import ModelConfig


def _gen(seed, name):
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _fp16_round(a):
    return a.astype(np.float16).astype(np.float32)


def _w(sd, seed, name, shape, fan_in, gain=1.0):
    # variance-preserving uniform init: var = gain^2 / fan_in
    bound = gain * np.sqrt(3.0 / fan_in)
    a = _gen(seed, name).uniform(-bound, bound, size=shape).astype(np.float32)
    sd[name] = torch.from_numpy(_fp16_round(a))


def _b(sd, seed, name, n, scale=0.05):
    a = (_gen(seed, name).standard_normal(n) * scale).astype(np.float32)
    sd[name] = torch.from_numpy(_fp16_round(a))


def _norm(sd, seed, pre, n):
    g = (1.0 + 0.1 * _gen(seed, pre + ".weight").standard_normal(n)).astype(np.float32)
    sd[pre + ".weight"] = torch.from_numpy(_fp16_round(g))
    _b(sd, seed, pre + ".bias", n, 0.1)


def _conv(sd, seed, pre, cin, cout, k, gain=1.0):
    _w(sd, seed, pre + ".weight", (cout, cin, k, k), cin * k * k, gain)
    _b(sd, seed, pre + ".bias", cout)


def _lin(sd, seed, pre, cin, cout, bias=True, gain=1.0):
    _w(sd, seed, pre + ".weight", (cout, cin), cin, gain)
    if bias:
        _b(sd, seed, pre + ".bias", cout)


def _resnet(sd, seed, pre, cin, cout, temb):
    _norm(sd, seed, pre + ".norm1", cin)
    _conv(sd, seed, pre + ".conv1", cin, cout, 3)
    if temb:
        _lin(sd, seed, pre + ".time_emb_proj", temb, cout)
    _norm(sd, seed, pre + ".norm2", cout)
    _conv(sd, seed, pre + ".conv2", cout, cout, 3, gain=0.7)
    if cin != cout:
        _conv(sd, seed, pre + ".conv_shortcut", cin, cout, 1)


def _transformer(sd, seed, pre, c, cross):
    _norm(sd, seed, pre + ".norm", c)
    _conv(sd, seed, pre + ".proj_in", c, c, 1)
    tb = pre + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        _norm(sd, seed, tb + "." + n, c)
    for a, kdim in (("attn1", c), ("attn2", cross)):
        _lin(sd, seed, tb + "." + a + ".to_q", c, c, bias=False, gain=1.2)
        _lin(sd, seed, tb + "." + a + ".to_k", kdim, c, bias=False, gain=1.2)
        _lin(sd, seed, tb + "." + a + ".to_v", kdim, c, bias=False)
        _lin(sd, seed, tb + "." + a + ".to_out.0", c, c, gain=0.7)
    _lin(sd, seed, tb + ".ff.net.0.proj", c, 8 * c)
    _lin(sd, seed, tb + ".ff.net.2", 4 * c, c, gain=0.7)
    _conv(sd, seed, pre + ".proj_out", c, c, 1, gain=0.7)


def unet_state_dict(cfg: ModelConfig, seed=0):
    sd = {}
    boc = cfg.block_out_channels
    n = len(boc)
    c0, te = boc[0], 4 * boc[0]
    _conv(sd, seed, "conv_in", cfg.in_channels, c0, 3)
    _lin(sd, seed, "time_embedding.linear_1", c0, te)
    _lin(sd, seed, "time_embedding.linear_2", te, te)
    out = c0
    for i in range(n):
        cin, out = out, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(sd, seed, "down_blocks.%d.resnets.%d" % (i, j), cin if j == 0 else out, out, te)
            if cfg.block_has_attn[i]:
                _transformer(sd, seed, "down_blocks.%d.attentions.%d" % (i, j), out, cfg.cross_dim)
        if i != n - 1:
            _conv(sd, seed, "down_blocks.%d.downsamplers.0.conv" % i, out, out, 3)
    cl = boc[-1]
    _resnet(sd, seed, "mid_block.resnets.0", cl, cl, te)
    _transformer(sd, seed, "mid_block.attentions.0", cl, cfg.cross_dim)
    _resnet(sd, seed, "mid_block.resnets.1", cl, cl, te)
    rev = list(reversed(boc))
    out = rev[0]
    for i in range(n):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, n - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin if j == cfg.layers_per_block else out
            rin = prev if j == 0 else out
            _resnet(sd, seed, "up_blocks.%d.resnets.%d" % (i, j), rin + skip, out, te)
            if cfg.block_has_attn[n - 1 - i]:
                _transformer(sd, seed, "up_blocks.%d.attentions.%d" % (i, j), out, cfg.cross_dim)
        if i != n - 1:
            _conv(sd, seed, "up_blocks.%d.upsamplers.0.conv" % i, out, out, 3)
    _norm(sd, seed, "conv_norm_out", c0)
    _conv(sd, seed, "conv_out", c0, cfg.out_channels, 3)
    return sd


def _vae_attn(sd, seed, pre, c):
    _norm(sd, seed, pre + ".group_norm", c)
    for n in ("query", "key"):
        _lin(sd, seed, pre + "." + n, c, c, gain=1.2)
    _lin(sd, seed, pre + ".value", c, c)
    _lin(sd, seed, pre + ".proj_attn", c, c, gain=0.7)


def vae_state_dict(cfg: ModelConfig, seed=0):
    sd = {}
    vb = cfg.vae_block_out_channels
    n = len(vb)
    L = cfg.vae_latent_channels
    _conv(sd, seed, "encoder.conv_in", cfg.vae_in_channels, vb[0], 3)
    out = vb[0]
    for i in range(n):
        cin, out = out, vb[i]
        for j in range(cfg.vae_layers_per_block):
            _resnet(sd, seed, "encoder.down_blocks.%d.resnets.%d" % (i, j), cin if j == 0 else out, out, 0)
        if i != n - 1:
            _conv(sd, seed, "encoder.down_blocks.%d.downsamplers.0.conv" % i, out, out, 3)
    cl = vb[-1]
    _resnet(sd, seed, "encoder.mid_block.resnets.0", cl, cl, 0)
    _vae_attn(sd, seed, "encoder.mid_block.attentions.0", cl)
    _resnet(sd, seed, "encoder.mid_block.resnets.1", cl, cl, 0)
    _norm(sd, seed, "encoder.conv_norm_out", cl)
    _conv(sd, seed, "encoder.conv_out", cl, 2 * L, 3)
    _conv(sd, seed, "quant_conv", 2 * L, 2 * L, 1)
    _conv(sd, seed, "post_quant_conv", L, L, 1)
    _conv(sd, seed, "decoder.conv_in", L, cl, 3)
    _resnet(sd, seed, "decoder.mid_block.resnets.0", cl, cl, 0)
    _vae_attn(sd, seed, "decoder.mid_block.attentions.0", cl)
    _resnet(sd, seed, "decoder.mid_block.resnets.1", cl, cl, 0)
    rev = list(reversed(vb))
    out = rev[0]
    for i in range(n):
        prev, out = out, rev[i]
        for j in range(cfg.vae_layers_per_block + 1):
            _resnet(sd, seed, "decoder.up_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else out, out, 0)
        if i != n - 1:
            _conv(sd, seed, "decoder.up_blocks.%d.upsamplers.0.conv" % i, out, out, 3)
    _norm(sd, seed, "decoder.conv_norm_out", vb[0])
    _conv(sd, seed, "decoder.conv_out", vb[0], cfg.vae_in_channels, 3)
    return sd


def synth_context(cfg: ModelConfig, rows, seed=0, name="context"):
    """Stand-in text-encoder output [rows, ctx_len, cross_dim] (fp16-representable), LayerNorm-like statistics."""
    a = _gen(seed, name).standard_normal((rows, cfg.ctx_len, cfg.cross_dim)).astype(np.float32)
    return torch.from_numpy(_fp16_round(a))


def clip_state_dict(cfg: ModelConfig, seed=0):
    """transformers CLIPTextModel state dict (key names of transformers >= 5; older releases prefix them with "text_model."),
    seeded, fp16-representable.  Scales follow the HF initialisation loosely; values only need to keep activations O(1)."""
    sd = {}
    H, I = cfg.cross_dim, cfg.clip_intermediate
    e = _gen(seed, "clip.tok").standard_normal((cfg.clip_vocab, H)).astype(np.float32) * 0.02
    sd["embeddings.token_embedding.weight"] = torch.from_numpy(_fp16_round(e))
    e = _gen(seed, "clip.pos").standard_normal((cfg.ctx_len, H)).astype(np.float32) * 0.02
    sd["embeddings.position_embedding.weight"] = torch.from_numpy(_fp16_round(e))
    for l in range(cfg.clip_layers):
        pre = "encoder.layers.%d" % l
        _norm(sd, seed, "clip." + pre + ".layer_norm1", H)
        _norm(sd, seed, "clip." + pre + ".layer_norm2", H)
        for n in ("q_proj", "k_proj", "v_proj"):
            _lin(sd, seed, "clip." + pre + ".self_attn." + n, H, H, gain=1.2)
        _lin(sd, seed, "clip." + pre + ".self_attn.out_proj", H, H, gain=0.5)
        _lin(sd, seed, "clip." + pre + ".mlp.fc1", H, I)
        _lin(sd, seed, "clip." + pre + ".mlp.fc2", I, H, gain=0.5)
    _norm(sd, seed, "clip.final_layer_norm", H)
    return {(k[5:] if k.startswith("clip.") else k): v for k, v in sd.items()}

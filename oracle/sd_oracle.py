"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

A plain-PyTorch (CPU, fp32 or fp64) restatement of the Stable-Diffusion-1.x UNet and VAE exactly as the reference executes
them through diffusers.  Parameters come from a diffusers-layout state dict.  Each function cites the in-tree statement of
the same graph it follows (/root/reference/models/edict/my_diffusers/models/...).

Pinned: oracle/make_golden.py runs the reference's own modules (my_diffusers UNet2DConditionModel / AutoencoderKL with the
reference's hooked attention, models/p2p/attention_control.py:20-47) on the same seeded weights and stores the outputs under
tests/golden/; tests/test_oracle_golden.py checks this file against them."""
import math

import torch
import torch.nn.functional as F


def _conv(sd, pre, x, stride=1, padding=1):
    return F.conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], stride=stride, padding=padding)


def _gn(sd, pre, x, groups, eps):
    return F.group_norm(x, groups, sd[pre + ".weight"], sd[pre + ".bias"], eps)


def _lin(sd, pre, x):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def timestep_embedding(t, dim, dtype):
    """get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)  (embeddings.py:21-60; computed in fp64)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float64) / half
    emb = torch.tensor([float(t)], dtype=torch.float64)[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)
    return emb.to(dtype)


def resnet(sd, pre, x, temb, groups, eps):
    """ResnetBlock2D.forward (resnet.py:331-365)."""
    h = F.silu(_gn(sd, pre + ".norm1", x, groups, eps))
    h = _conv(sd, pre + ".conv1", h)
    if temb is not None:
        h = h + _lin(sd, pre + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(_gn(sd, pre + ".norm2", h, groups, eps))
    h = _conv(sd, pre + ".conv2", h)
    if pre + ".conv_shortcut.weight" in sd:
        x = _conv(sd, pre + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd, pre, x, context, heads, hook, place):
    """CrossAttention as hooked by register_attention_control (attention_control.py:20-47; attention.py:236-288)."""
    is_cross = context is not None
    ctx = context if is_cross else x
    q, k, v = _lin(sd, pre + ".to_q", x), _lin(sd, pre + ".to_k", ctx), _lin(sd, pre + ".to_v", ctx)
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)

    q, k, v = split(q), split(k), split(v)
    scale = d ** -0.5
    sim = torch.einsum("bid,bjd->bij", q, k) * scale
    attn = sim.softmax(dim=-1)
    if hook is not None and hasattr(hook, "qkv_editor"):
        # MasaCtrl-style hook (masactrl_utils.py:85-127): the editor gets q, k, v as well and returns the merged-head output
        out = hook.qkv_editor(q, k, v, sim, attn, is_cross, place, heads, scale)
        return _lin(sd, pre + ".to_out.0", out)
    if hook is not None:
        attn = hook(attn, is_cross, place)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    return _lin(sd, pre + ".to_out.0", out)


def transformer(sd, pre, x, context, heads, groups, hook, place):
    """SpatialTransformer + BasicTransformerBlock + GEGLU FeedForward (attention.py:140-151, 192-200, 303-333)."""
    b, c, h, w = x.shape
    x_in = x
    x = _gn(sd, pre + ".norm", x, groups, 1e-6)
    x = _conv(sd, pre + ".proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    tb = pre + ".transformer_blocks.0"
    ln = lambda name, t: F.layer_norm(t, (c,), sd[tb + "." + name + ".weight"], sd[tb + "." + name + ".bias"], 1e-5)
    x = attention(sd, tb + ".attn1", ln("norm1", x), None, heads, hook, place) + x
    x = attention(sd, tb + ".attn2", ln("norm2", x), context, heads, hook, place) + x
    hgate = _lin(sd, tb + ".ff.net.0.proj", ln("norm3", x))
    a, gate = hgate.chunk(2, dim=-1)
    x = _lin(sd, tb + ".ff.net.2", a * F.gelu(gate)) + x
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = _conv(sd, pre + ".proj_out", x, padding=0)
    return x + x_in


def unet_forward(sd, cfg, sample, t, context, hook=None):
    """UNet2DConditionModel.forward (unet_2d_condition.py:189-273).  `hook(attn, is_cross, place)` is called at each of
    the 32 attention sites in the order down -> mid -> up, self before cross (Appendix B of SURVEY.md)."""
    boc = cfg.block_out_channels
    n = len(boc)
    G, eps = cfg.norm_groups, 1e-5
    dtype = sample.dtype
    temb = timestep_embedding(t, boc[0], dtype).expand(sample.shape[0], -1)
    temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", temb)))
    h = _conv(sd, "conv_in", sample)
    skips = [h]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet(sd, "down_blocks.%d.resnets.%d" % (i, j), h, temb, G, eps)
            if cfg.block_has_attn[i]:
                h = transformer(sd, "down_blocks.%d.attentions.%d" % (i, j), h, context, cfg.heads, G, hook, "down")
            skips.append(h)
        if i != n - 1:
            h = _conv(sd, "down_blocks.%d.downsamplers.0.conv" % i, h, stride=2, padding=1)
            skips.append(h)
    h = resnet(sd, "mid_block.resnets.0", h, temb, G, eps)
    h = transformer(sd, "mid_block.attentions.0", h, context, cfg.heads, G, hook, "mid")
    h = resnet(sd, "mid_block.resnets.1", h, temb, G, eps)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, "up_blocks.%d.resnets.%d" % (i, j), h, temb, G, eps)
            if cfg.block_has_attn[n - 1 - i]:
                h = transformer(sd, "up_blocks.%d.attentions.%d" % (i, j), h, context, cfg.heads, G, hook, "up")
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, "up_blocks.%d.upsamplers.0.conv" % i, h)
    h = F.silu(_gn(sd, "conv_norm_out", h, G, eps))
    return _conv(sd, "conv_out", h)


def vae_attention(sd, pre, x, groups):
    """AttentionBlock.forward, one head (attention.py:54-92)."""
    b, c, hh, ww = x.shape
    r = x
    h = _gn(sd, pre + ".group_norm", x, groups, 1e-6)
    h = h.view(b, c, hh * ww).transpose(1, 2)
    q, k, v = _lin(sd, pre + ".query", h), _lin(sd, pre + ".key", h), _lin(sd, pre + ".value", h)
    scale = 1 / math.sqrt(math.sqrt(c))
    p = torch.softmax(torch.matmul(q * scale, k.transpose(-1, -2) * scale), dim=-1)
    h = torch.matmul(p, v)
    h = _lin(sd, pre + ".proj_attn", h)
    h = h.transpose(-1, -2).reshape(b, c, hh, ww)
    return h + r


def vae_encode_mean(sd, cfg, x):
    """AutoencoderKL.encode(x).latent_dist.mean (vae.py:113-130, 552-560, 329-336); Downsample2D padding=0 (resnet.py:89-95)."""
    vb = cfg.vae_block_out_channels
    n = len(vb)
    G, eps = cfg.vae_norm_groups, 1e-6
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(n):
        for j in range(cfg.vae_layers_per_block):
            h = resnet(sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), h, None, G, eps)
        if i != n - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, h, stride=2, padding=0)
    h = resnet(sd, "encoder.mid_block.resnets.0", h, None, G, eps)
    h = vae_attention(sd, "encoder.mid_block.attentions.0", h, G)
    h = resnet(sd, "encoder.mid_block.resnets.1", h, None, G, eps)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, G, eps))
    h = _conv(sd, "encoder.conv_out", h)
    moments = _conv(sd, "quant_conv", h, padding=0)
    return moments[:, : cfg.vae_latent_channels]


def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode(z).sample (vae.py:191-209, 562-566)."""
    vb = cfg.vae_block_out_channels
    n = len(vb)
    G, eps = cfg.vae_norm_groups, 1e-6
    h = _conv(sd, "post_quant_conv", z, padding=0)
    h = _conv(sd, "decoder.conv_in", h)
    h = resnet(sd, "decoder.mid_block.resnets.0", h, None, G, eps)
    h = vae_attention(sd, "decoder.mid_block.attentions.0", h, G)
    h = resnet(sd, "decoder.mid_block.resnets.1", h, None, G, eps)
    for i in range(n):
        for j in range(cfg.vae_layers_per_block + 1):
            h = resnet(sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), h, None, G, eps)
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, "decoder.up_blocks.%d.upsamplers.0.conv" % i, h)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, G, eps))
    return _conv(sd, "decoder.conv_out", h)


def cast_sd(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------- CLIP text encoder
def clip_text_forward(sd, cfg, input_ids):
    """transformers CLIPTextModel(input_ids)[0] = last_hidden_state -- what `model.text_encoder(ids)[0]` is in the reference
    (models/p2p/inversion.py:290-306, p2p_guidance_forward.py:151-164).  transformers is a third-party dependency of the reference
    (environment/p2p_requirements.txt:2, unpinned; absent from /root/reference); this restates its published CLIPTextTransformer:
    token + position embeddings; L pre-LayerNorm blocks [causal multi-head self-attention, quick_gelu MLP]; final LayerNorm.
    Pinned against the installed transformers (5.15) in tests/golden/clip_*.npz.  sd: CLIPTextModel state dict, keys without the
    older "text_model." prefix."""
    F = torch.nn.functional
    H, heads = cfg.cross_dim, cfg.clip_heads
    d = H // heads
    ids = torch.as_tensor(input_ids).long()
    B, T = ids.shape
    x = sd["embeddings.token_embedding.weight"].float()[ids] + sd["embeddings.position_embedding.weight"].float()[None, :T]
    mask = torch.full((T, T), float("-inf")).triu(1)
    for l in range(cfg.clip_layers):
        p = "encoder.layers.%d." % l
        ln = lambda n, t: F.layer_norm(t, (H,), sd[p + n + ".weight"].float(), sd[p + n + ".bias"].float(), 1e-5)
        lin = lambda n, t: F.linear(t, sd[p + n + ".weight"].float(), sd[p + n + ".bias"].float())
        h = ln("layer_norm1", x)
        q, k, v = (lin("self_attn." + n, h).reshape(B, T, heads, d).transpose(1, 2) for n in ("q_proj", "k_proj", "v_proj"))
        att = (q @ k.transpose(-1, -2)) * d ** -0.5 + mask
        o = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H)
        x = x + lin("self_attn.out_proj", o)
        h = lin("mlp.fc1", ln("layer_norm2", x))
        x = x + lin("mlp.fc2", h * torch.sigmoid(1.702 * h))
    return F.layer_norm(x, (H,), sd["final_layer_norm.weight"].float(), sd["final_layer_norm.bias"].float(), 1e-5)

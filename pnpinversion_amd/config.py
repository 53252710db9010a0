"""Architecture description shared by the native pipeline, the weight generator and the tests.

SD-1.x values: diffusers ctor args of UNet2DConditionModel / AutoencoderKL as used by the reference
(models/edict/my_diffusers/models/unet_2d_condition.py:57-82, vae.py:508-519) and the in-tree v1-inference.yaml
(models/instructpix2pix/stable_diffusion/configs/stable-diffusion/v1-inference.yaml:29-65)."""
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class ModelConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    block_has_attn: Tuple[int, ...] = (1, 1, 1, 0)
    layers_per_block: int = 2
    heads: int = 8
    cross_dim: int = 768
    ctx_len: int = 77
    sample_size: int = 64
    norm_groups: int = 32
    n_train_timesteps: int = 1000
    vae_in_channels: int = 3
    vae_latent_channels: int = 4
    vae_block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    vae_layers_per_block: int = 2
    vae_norm_groups: int = 32
    # CLIP text encoder (transformers CLIPTextModel of SD-1.x: ViT-L/14 text tower); hidden = cross_dim, positions = ctx_len
    clip_layers: int = 12
    clip_heads: int = 12
    clip_intermediate: int = 3072
    clip_vocab: int = 49408

    @property
    def n_blocks(self):
        return len(self.block_out_channels)

    @property
    def vae_scale(self):
        return 2 ** (len(self.vae_block_out_channels) - 1)

    def to_c(self):
        from . import _capi
        c = _capi.ModelConfig()
        c.in_channels, c.out_channels, c.n_blocks = self.in_channels, self.out_channels, self.n_blocks
        for i in range(self.n_blocks):
            c.block_out_channels[i] = self.block_out_channels[i]
            c.block_has_attn[i] = self.block_has_attn[i]
        c.layers_per_block, c.heads, c.cross_dim, c.ctx_len = self.layers_per_block, self.heads, self.cross_dim, self.ctx_len
        c.sample_size, c.norm_groups, c.n_train_timesteps = self.sample_size, self.norm_groups, self.n_train_timesteps
        c.vae_in_channels, c.vae_latent_channels = self.vae_in_channels, self.vae_latent_channels
        c.vae_n_blocks = len(self.vae_block_out_channels)
        for i in range(c.vae_n_blocks):
            c.vae_block_out_channels[i] = self.vae_block_out_channels[i]
        c.vae_layers_per_block, c.vae_norm_groups = self.vae_layers_per_block, self.vae_norm_groups
        c.clip_layers, c.clip_heads = self.clip_layers, self.clip_heads
        c.clip_intermediate, c.clip_vocab = self.clip_intermediate, self.clip_vocab
        return c


SD1 = ModelConfig()

# Reduced-width configurations used by the parity tests (same graph, same code paths, CPU-oracle friendly).
# (layers_per_block stays 2: the 0.3.0 fork the oracle is pinned against mis-sizes its downsampler for 1-layer blocks)
TINY16 = ModelConfig(block_out_channels=(32, 64, 64, 64), cross_dim=64, sample_size=16, layers_per_block=2,
                     vae_block_out_channels=(32, 32, 64, 64), vae_layers_per_block=2, clip_layers=2, clip_heads=2, clip_intermediate=256)
# 64x64 latents (needed by LocalBlend's hard-coded 16x16 maps) with narrow channels
SMALL64 = ModelConfig(block_out_channels=(32, 64, 128, 128), cross_dim=64, sample_size=64, layers_per_block=2,
                      vae_block_out_channels=(32, 32, 64, 64), vae_layers_per_block=2, clip_layers=2, clip_heads=2, clip_intermediate=256)
# SMALL64 without attention at the 64 x 64 level: LocalBlend's five 16 x 16 cross-attention maps (down_cross[2:4] + up_cross[:3] of the
# stored <= 32^2-token maps) index the same layers, but the CPU oracle no longer materialises 4096 x 4096 self-attention tensors --
# used for the full 50 + 50-step schedule against the oracle (22 attention sites instead of 32)
SMALL64_LB = ModelConfig(block_out_channels=(32, 64, 128, 128), block_has_attn=(0, 1, 1, 0), cross_dim=64, sample_size=64, layers_per_block=2,
                         vae_block_out_channels=(32, 32, 64, 64), vae_layers_per_block=2, clip_layers=2, clip_heads=2, clip_intermediate=256)

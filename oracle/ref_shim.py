"""TEST INFRASTRUCTURE ONLY.  Makes the reference's own Python for the hot path importable in the build container
(/root/reference is read-only and absent on the GPU box; nothing under tests -m gpu / smoke() / bench.py imports this).

Recipe:
  * the in-tree diffusers 0.3.0 fork models/edict/my_diffusers is imported without running its package __init__
    (which needs the real, absent `diffusers`);
  * a minimal stand-in `diffusers` module provides StableDiffusionPipeline (placeholder) and a DDIMScheduler base class
    restating the 0.10.0 tables (fp32 scaled_linear betas, set_timesteps) -- DDIMSchedulerDev.step itself is the reference's;
  * utils.utils.txt_draw / models.p2p_editor.txt_draw are stubbed (matplotlib / numpy API drift, not on the numeric path);
  * tensors the reference pins to "cuda" (attention_control.py:291,307,326,355) are redirected to the CPU.
No reference source is copied: modules are imported from where they lie."""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("PNPI_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "models", "p2p"))


class _Config(dict):
    __getattr__ = dict.__getitem__


def _install_fake_diffusers():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_pnpi_fake", False):
        return
    dm = types.ModuleType("diffusers")
    dm._pnpi_fake = True
    dm.__version__ = "0.10.0"

    class StableDiffusionPipeline:  # placeholder; the oracle builds P2PEditor without from_pretrained
        pass

    class DDIMSchedulerOutput(dict):
        def __init__(self, prev_sample=None, pred_original_sample=None):
            super().__init__(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
            self.prev_sample = prev_sample
            self.pred_original_sample = pred_original_sample

    class DDIMScheduler:
        def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon"):
            assert beta_schedule == "scaled_linear"
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
            self.alphas = 1.0 - self.betas
            self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
            self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
            self.init_noise_sigma = 1.0
            self.num_inference_steps = None
            self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
            self.config = _Config(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                                  steps_offset=steps_offset, clip_sample=clip_sample)

        def set_timesteps(self, num_inference_steps, device=None):
            self.num_inference_steps = num_inference_steps
            ratio = self.config.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    dm.StableDiffusionPipeline = StableDiffusionPipeline
    dm.DDIMScheduler = DDIMScheduler
    sched = types.ModuleType("diffusers.schedulers")
    sd = types.ModuleType("diffusers.schedulers.scheduling_ddim")
    sd.DDIMScheduler = DDIMScheduler
    sd.DDIMSchedulerOutput = DDIMSchedulerOutput
    sched.scheduling_ddim = sd
    dm.schedulers = sched
    sys.modules["diffusers"] = dm
    sys.modules["diffusers.schedulers"] = sched
    sys.modules["diffusers.schedulers.scheduling_ddim"] = sd


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF)
    pkg = types.ModuleType("my_diffusers")
    pkg.__path__ = [os.path.join(REF, "models", "edict", "my_diffusers")]
    pkg.__version__ = "0.3.0"
    sys.modules["my_diffusers"] = pkg
    _install_fake_diffusers()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    import utils.utils as ru

    def txt_draw(text, target_size=(512, 512)):
        return np.full((target_size[0], target_size[1], 3), 255, dtype=np.uint8)

    ru.txt_draw = txt_draw
    import models.p2p_editor as pe
    pe.txt_draw = txt_draw
    _installed = True


def ref_models():
    install()
    from my_diffusers.models.unet_2d_condition import UNet2DConditionModel
    from my_diffusers.models.vae import AutoencoderKL
    return UNet2DConditionModel, AutoencoderKL


def build_unet(cfg, sd, dtype=torch.float32):
    UNet, _ = ref_models()
    n = len(cfg.block_out_channels)
    down = tuple("CrossAttnDownBlock2D" if cfg.block_has_attn[i] else "DownBlock2D" for i in range(n))
    up = tuple("CrossAttnUpBlock2D" if cfg.block_has_attn[n - 1 - i] else "UpBlock2D" for i in range(n))
    m = UNet(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
             layers_per_block=cfg.layers_per_block, block_out_channels=tuple(cfg.block_out_channels),
             down_block_types=down, up_block_types=up, cross_attention_dim=cfg.cross_dim, attention_head_dim=cfg.heads)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # Downsample2D registers its conv twice (conv / Conv2d_0): only that alias may be missing from our key set
    assert all(".Conv2d_0." in k for k in missing), missing
    assert not unexpected, unexpected
    m = m.to(dtype).eval()
    if dtype == torch.float32:
        m.conv_norm_out.double()   # fork quirk: unet_2d_condition.py:266 feeds sample.double() to this GroupNorm
    m.in_channels = cfg.in_channels
    return m


def build_vae(cfg, sd, dtype=torch.float32):
    _, VAE = ref_models()
    n = len(cfg.vae_block_out_channels)
    m = VAE(in_channels=cfg.vae_in_channels, out_channels=cfg.vae_in_channels, down_block_types=("DownEncoderBlock2D",) * n,
            up_block_types=("UpDecoderBlock2D",) * n, block_out_channels=tuple(cfg.vae_block_out_channels),
            layers_per_block=cfg.vae_layers_per_block, latent_channels=cfg.vae_latent_channels,
            sample_size=cfg.sample_size * cfg.vae_scale)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert all(".Conv2d_0." in k for k in missing), missing
    assert not unexpected, unexpected
    return m.to(dtype).eval()


class _Holder:
    pass


class cuda_to_cpu:
    """Context manager: Tensor.to("cuda") -> CPU while the reference builds its controllers on a GPU-less host."""

    def __enter__(self):
        self._orig = torch.Tensor.to

        def to(t, *a, **k):
            a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
            if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                k["device"] = "cpu"
            return self._orig(t, *a, **k)

        torch.Tensor.to = to
        return self

    def __exit__(self, *exc):
        torch.Tensor.to = self._orig


def build_editor(cfg, unet_sd, vae_sd, tokenizer, text_encoder, num_ddim_steps, dtype=torch.float32):
    """The reference's P2PEditor (models/p2p_editor.py:12-25) on CPU with seeded weights, skipping from_pretrained."""
    install()
    import models.p2p_editor as pe
    from models.p2p.scheduler_dev import DDIMSchedulerDev
    ed = pe.P2PEditor.__new__(pe.P2PEditor)
    ed.device = torch.device("cpu")
    ed.method_list = ["directinversion+p2p"]
    ed.num_ddim_steps = num_ddim_steps
    ed.scheduler = DDIMSchedulerDev(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                    set_alpha_to_one=False)
    h = _Holder()
    h.unet = build_unet(cfg, unet_sd, dtype)
    h.vae = build_vae(cfg, vae_sd, dtype)
    h.tokenizer = tokenizer
    h.text_encoder = text_encoder
    h.scheduler = ed.scheduler
    h.device = torch.device("cpu")
    ed.ldm_stable = h
    ed.scheduler.set_timesteps(num_ddim_steps)
    return ed


# ---------------------------------------------------------------------------------------------------------- MasaCtrl
def _install_fake_vision():
    """run_editing_masactrl.py / models/masactrl import torchvision (read_image, save_image) and cv2, neither installed here.
    read_image is the only one the editing path calls: PIL stands in (uint8 CHW tensor, as torchvision returns)."""
    if "torchvision" in sys.modules and getattr(sys.modules["torchvision"], "_pnpi_fake", False):
        return
    tv = types.ModuleType("torchvision")
    tv._pnpi_fake = True
    io = types.ModuleType("torchvision.io")
    ut = types.ModuleType("torchvision.utils")

    def read_image(path):
        from PIL import Image
        arr = np.array(Image.open(path).convert("RGB")) if isinstance(path, str) else np.asarray(path)
        return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)

    io.read_image = read_image
    ut.save_image = lambda *a, **k: None
    tv.io, tv.utils = io, ut
    sys.modules["torchvision"], sys.modules["torchvision.io"], sys.modules["torchvision.utils"] = tv, io, ut
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")


class _UNetOut(dict):
    """my_diffusers returns {"sample": ...}; the MasaCtrl pipeline (diffusers 0.15 API) reads `.sample`."""
    __getattr__ = dict.__getitem__


def build_masactrl_editor(cfg, unet_sd, vae_sd, tokenizer, text_encoder, num_ddim_steps):
    """The reference's MasaCtrlEditor (run_editing_masactrl.py:57-74) on CPU with seeded weights, skipping from_pretrained.
    The fork's attention class is called CrossAttention; the MasaCtrl hook looks for the diffusers-0.15 name `Attention`
    (masactrl_utils.py:130), so the modules get a renamed subclass (same code)."""
    install()
    _install_fake_vision()
    import diffusers
    import run_editing_masactrl as rem
    from models.masactrl.diffuser_utils import MasaCtrlPipeline
    import utils.utils as ru
    rem.txt_draw = ru.txt_draw
    ed = rem.MasaCtrlEditor.__new__(rem.MasaCtrlEditor)
    ed.device = torch.device("cpu")
    ed.method_list = ["directinversion+masactrl"]
    ed.num_ddim_steps = num_ddim_steps
    ed.scheduler = diffusers.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                           set_alpha_to_one=False)
    unet = build_unet(cfg, unet_sd)
    for m in unet.modules():
        if m.__class__.__name__ == "CrossAttention":
            m.__class__ = type("Attention", (m.__class__,), {})
    fwd = unet.forward
    unet.forward = lambda *a, **k: _UNetOut(fwd(*a, **k))
    pipe = MasaCtrlPipeline.__new__(MasaCtrlPipeline)
    pipe.unet, pipe.vae = unet, build_vae(cfg, vae_sd)
    pipe.tokenizer, pipe.text_encoder, pipe.scheduler = tokenizer, text_encoder, ed.scheduler
    pipe.device = torch.device("cpu")
    ed.model = pipe
    ed.scheduler.set_timesteps(num_ddim_steps)
    return ed

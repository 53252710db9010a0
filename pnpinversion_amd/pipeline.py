"""NativePipeline: the duck-type the reference's P2P code expects from StableDiffusionPipeline
(models/p2p_editor.py:23-25: `.unet`, `.vae`, `.scheduler`, `.tokenizer`, `.text_encoder`, `.device`), backed by libpnpi.

  model.unet(latents, t, encoder_hidden_states=ctx)["sample"]           inversion.py:273, p2p_guidance_forward.py:109
  model.unet.in_channels                                                 utils/utils.py:51,54
  model.vae.encode(x)['latent_dist'].mean / model.vae.decode(z)['sample'] utils/utils.py:61,78
  model.scheduler.{timesteps, alphas_cumprod, final_alpha_cumprod, config, num_inference_steps, set_timesteps, step}
"""
import numpy as np
import torch

from .config import SD1, ModelConfig
from .engine import NativeEngine
from .p2p.scheduler_dev import DDIMSchedulerDev
from .text import SyntheticTextEncoder, WordTokenizer


class _Dist:
    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean


class NativeVAE:
    def __init__(self, engine: NativeEngine):
        self.engine = engine
        self.device = engine.device

    def encode(self, x):
        return {"latent_dist": _Dist(self.engine.vae_encode(x))}

    def decode(self, z):
        return {"sample": self.engine.vae_decode(z)}

    # fused uint8 paths used by utils.latent2image / image2latent
    def latent2image_u8(self, latents):
        latents, m = latents.detach(), self.engine.max_vae_images       # the VAE workspace holds max_vae_images images
        return np.concatenate([self.engine.latent2image(latents[i:i + m]).cpu().numpy() for i in range(0, latents.shape[0], m)])

    def image2latent_u8(self, image):
        image, m = np.ascontiguousarray(image), self.engine.max_vae_images
        if image.ndim == 3:
            return self.engine.image2latent(image)
        return torch.cat([self.engine.image2latent(image[i:i + m]) for i in range(0, image.shape[0], m)])


class _NotExecuted:
    """to_q / to_k / to_v / to_out of an attention-site marker: the projections run inside libpnpi, never through Python."""

    def __call__(self, *a, **k):
        raise RuntimeError("the native UNet executes its attention sites inside libpnpi; this marker only lets "
                           "register_attention_control find and hook the site")


class CrossAttention:
    """Marker object for ONE of the UNet's attention sites.  The reference's unmodified register_attention_control
    (models/p2p/attention_control.py:12-81) walks `model.unet.named_children()`, recognises modules by this CLASS NAME (:63) and
    assigns each a new `.forward` closure over the controller (:64).  The assignment is what the native UNet listens for: it reads
    the controller out of the closure and routes the whole UNet through it (kernel descriptor when the controller's edit is one the
    library knows, the materialise-and-call-back path of pnpi_set_attention_callback otherwise)."""

    def __init__(self, unet, index, place, is_cross, heads, dim_head):
        d = self.__dict__
        d["_unet"], d["index"], d["place_in_unet"], d["is_cross"] = unet, index, place, is_cross
        d["heads"], d["scale"] = heads, dim_head ** -0.5
        d["to_q"] = d["to_k"] = d["to_v"] = d["to_out"] = _NotExecuted()

    def children(self):
        return iter(())

    def reshape_heads_to_batch_dim(self, t):
        raise RuntimeError("attention-site marker: not executed in Python")

    reshape_batch_dim_to_heads = reshape_heads_to_batch_dim

    def __setattr__(self, name, value):
        if name == "forward":
            self._unet._site_hooked(self, value)
        self.__dict__[name] = value


class _SiteContainer:
    """down / mid / up container of attention-site markers (what `named_children()` yields)."""

    def __init__(self, sites):
        self._sites = list(sites)

    def children(self):
        return iter(self._sites)


def _controller_of_hook(fn):
    """The controller a hooked `forward` closes over (attention_control.py:20-47: free variable `controller`)."""
    code, cells = getattr(fn, "__code__", None), getattr(fn, "__closure__", None)
    if code is not None and cells:
        for name, cell in zip(code.co_freevars, cells):
            if name == "controller":
                return cell.cell_contents
    raise TypeError("an attention site of the native UNet was given a forward hook that does not close over a `controller` "
                    "(the reference's register_attention_control protocol): the native UNet cannot run arbitrary Python attention")


class NativeUNet:
    """Callable with the reference's UNet protocol.  A registered controller (see p2p.attention_control.register_attention_control,
    or the reference's own unmodified function through the attention-site markers of named_children()) is translated into the kernel
    descriptor; it applies to batches laid out [uncond_src, uncond_tgt, cond_src, cond_tgt]."""

    def __init__(self, engine: NativeEngine):
        self.engine = engine
        self.in_channels = engine.cfg.in_channels
        self.controller = None
        cfg = engine.cfg
        n_attn_blocks = sum(cfg.block_has_attn)
        self.num_att_layers = 2 * (cfg.layers_per_block * n_attn_blocks + 1 + (cfg.layers_per_block + 1) * n_attn_blocks)
        # attention-site markers in the order the forward visits them: per transformer block attn1 (self), attn2 (cross)
        heads = cfg.heads
        sites = {"down": [], "mid": [], "up": []}

        def add(place, channels, count):
            for _ in range(count):
                for is_cross in (False, True):
                    sites[place].append(CrossAttention(self, sum(len(v) for v in sites.values()), place, is_cross, heads, channels // heads))
        boc = list(cfg.block_out_channels)
        for i, has in enumerate(cfg.block_has_attn):
            if has:
                add("down", boc[i], cfg.layers_per_block)
        add("mid", boc[-1], 1)
        for i, has in reversed(list(enumerate(cfg.block_has_attn))):
            if has:
                add("up", boc[i], cfg.layers_per_block + 1)
        self._sites = sites
        assert sum(len(v) for v in sites.values()) == self.num_att_layers

    def set_controller(self, controller):
        from .p2p.attention_control import is_callback_controller
        self.controller = controller
        if controller is not None and getattr(controller, "local_blend", None) is not None:
            # per-forward (level-1) drivers call controller.step_callback(latents): LocalBlend then runs on this engine's accumulators
            try:
                controller._pnpi_engine = self.engine
            except AttributeError:
                pass
        if not is_callback_controller(controller) and getattr(self, "_cb_for", None) is not None:
            # a host callback left by an earlier callback controller must not outlive it: the loop entry points (level 2) and the next
            # descriptor forward would otherwise run the stale Python controller
            self.engine.set_attention_callback(None)
            self._cb_for = None

    def named_children(self):
        """(name, container) pairs whose names contain "down" / "mid" / "up" (attention_control.py:72-79) and whose children are the
        attention-site markers the reference's register_attention_control hooks."""
        return iter([("down_blocks", _SiteContainer(self._sites["down"])), ("mid_block", _SiteContainer(self._sites["mid"])),
                     ("up_blocks", _SiteContainer(self._sites["up"]))])

    def _site_hooked(self, site, fn):
        """An attention-site marker was assigned a `.forward` (the reference's register_attention_control, attention_control.py:64): every
        site of one registration closes over the same controller, and each registration starts again at the first site."""
        from .p2p.attention_control import adapt_foreign_controller
        raw = _controller_of_hook(fn)
        if site.index == 0 or getattr(self, "_hooked_raw", None) is not raw:
            self._hooked_raw = raw
            self.set_controller(adapt_foreign_controller(raw))

    def __call__(self, sample, timestep, encoder_hidden_states=None, **kw):
        rows = sample.shape[0]
        t = int(timestep)
        ctrls, cur_step, rpi = None, 0, 1
        c = self.controller
        from .p2p.attention_control import controller_tables, is_callback_controller
        if is_callback_controller(c):
            # level-1 fallback: an object of the reference's controller protocol without a kernel descriptor is called back
            # on the materialised probabilities at each of the 32 attention sites, exactly as the reference's hooked forward does
            # (attention_control.py:40-44).  The controller advances its own cur_step / cur_att_layer, as in the reference.
            if getattr(self, "_cb_for", None) is not c:
                self.engine.set_attention_callback(lambda attn, is_cross, place, layer: c(attn, is_cross, place))
                self._cb_for = c
            return {"sample": self.engine.unet(sample, t, encoder_hidden_states)}
        if getattr(self, "_cb_for", None) is not None:
            self.engine.set_attention_callback(None)
            self._cb_for = None
        tables = controller_tables(c)        # raises for a controller object the library has no descriptor for
        if tables is not None:
            if rows % 4 != 0:
                raise ValueError("an attention controller is registered: the UNet batch must be [uncond_src, uncond_tgt, cond_src, "
                                 "cond_tgt] per image (rows % 4 == 0), got %d rows" % rows)
            ctrls, cur_step, rpi = [tables] * (rows // 4), c.cur_step, 4
        eps = self.engine.unet(sample, t, encoder_hidden_states, rows_per_image=rpi, ctrls=ctrls, cur_step=cur_step)
        if c is not None and hasattr(c, "cur_step"):
            # the reference advances cur_step after the 32nd attention site of every UNet call (attention_control.py:186-189)
            c.cur_step += 1
            c.between_steps()
        return {"sample": eps}


class NativeTextEncoder:
    """`model.text_encoder(input_ids)[0]` (inversion.py:290-306) on the device: the CLIP text transformer of libpnpi."""

    def __init__(self, engine: NativeEngine):
        self.engine = engine
        self.device = engine.device

    def to(self, device):
        return self

    def __call__(self, input_ids, **kw):
        return (self.engine.text_encode(input_ids),)


class NativePipeline:
    def __init__(self, cfg: ModelConfig = SD1, device=None, max_unet_rows=12, max_vae_images=2, tokenizer=None, text_encoder=None,
                 scheduler=None, share_weights_with=None):
        """text_encoder: None = seeded stand-in embedding (no CLIP weights needed); "native" = the CLIP text transformer of
        libpnpi (load its weights with load_state_dict(..., clip_sd=...)); or any callable with the CLIPTextModel protocol.
        share_weights_with: another NativePipeline -- a further context (own stream / workspaces) on ITS packed weights, no copy."""
        # what a further context on the same weights is built from (P2PEditor's in-flight / stage-overlap peers): every constructor option
        self._ctor = dict(cfg=cfg, device=device, max_unet_rows=max_unet_rows, max_vae_images=max_vae_images, tokenizer=tokenizer,
                          text_encoder=text_encoder)
        self.engine = NativeEngine(cfg, device=device, max_unet_rows=max_unet_rows, max_vae_images=max_vae_images,
                                   share_weights_with=share_weights_with.engine if share_weights_with is not None else None)
        self.device = self.engine.device
        self.unet = NativeUNet(self.engine)
        self.vae = NativeVAE(self.engine)
        self.tokenizer = tokenizer or WordTokenizer()
        if isinstance(text_encoder, str) and text_encoder == "native":
            text_encoder = NativeTextEncoder(self.engine)
        self.text_encoder = text_encoder or SyntheticTextEncoder(cfg.cross_dim, device=self.device)
        self.scheduler = scheduler or DDIMSchedulerDev(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                                       clip_sample=False, set_alpha_to_one=False)
        self.scheduler.bind(self.engine)

    def to(self, device):
        return self

    def peer(self, **override):
        """A further pipeline of the same class on this one's packed weights (pnpi_create_shared: own HIP stream = torch's current
        stream, own workspaces), built from this pipeline's own constructor options and a copy of its scheduler (same alphas, bound to
        the new context); `override` replaces some (e.g. max_unet_rows)."""
        import copy
        kw = dict(self._ctor)
        kw["device"] = self.device
        kw["scheduler"] = copy.copy(self.scheduler)
        kw.update(override)
        return type(self)(share_weights_with=self, **kw)

    def load_state_dict(self, unet_sd, vae_sd, clip_sd=None):
        self.engine.load_state_dict(unet_sd, vae_sd, clip_sd=clip_sd)
        n, names = self.engine.missing_weights()
        if n:
            raise RuntimeError("missing %d weight tensors, e.g. %s" % (n, names[:5]))

    @classmethod
    def synthetic(cls, cfg: ModelConfig = SD1, seed=0, **kw):
        """Seeded random-weight pipeline (no checkpoints exist on the build / GPU boxes)."""
        from . import weights
        p = cls(cfg, **kw)
        clip = weights.clip_state_dict(cfg, seed) if isinstance(p.text_encoder, NativeTextEncoder) else None
        p.load_state_dict(weights.unet_state_dict(cfg, seed), weights.vae_state_dict(cfg, seed), clip_sd=clip)
        return p

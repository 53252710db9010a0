"""NativeEngine: thin Python owner of one pnpi_ctx (one per process / GPU).  PyTorch is used for device memory, the HIP
stream and (elsewhere) torch.distributed only; every FLOP of the hot path runs inside libpnpi.so."""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .config import ModelConfig


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class ControllerTables:
    """Host tables of one image's Prompt-to-Prompt controller -> pnpi_ctrl_desc (see include/pnpi.h)."""

    def __init__(self, cross_alpha, mapper, alphas, equalizer, self_range, lb_alpha=None, lb_start=0, lb_threshold=0.3,
                 self_max_tokens=32 ** 2, lb_sub_alpha=None, lb_threshold_sub=0.3):
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        self.cross_alpha = f(cross_alpha)      # [steps+1, 77]
        self.mapper = f(mapper)                # [77, 77]
        self.alphas = f(alphas)                # [77]
        self.equalizer = f(equalizer)          # [77]
        self.self_range = (int(self_range[0]), int(self_range[1]))
        self.lb_alpha = f(lb_alpha) if lb_alpha is not None else None   # [2, 77]
        self.lb_start = int(lb_start)
        self.lb_threshold = float(lb_threshold)
        self.self_max_tokens = int(self_max_tokens)
        self.lb_sub_alpha = f(lb_sub_alpha) if lb_sub_alpha is not None else None   # [2, 77] LocalBlend substruct_layers
        self.lb_threshold_sub = float(lb_threshold_sub)
        if self.lb_sub_alpha is not None and self.lb_alpha is None:
            raise ValueError("lb_sub_alpha (LocalBlend substruct_words) without lb_alpha (LocalBlend words)")

    def desc(self):
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        d = _capi.CtrlDesc()
        d.kind = 1
        d.n_alpha_rows = self.cross_alpha.shape[0]
        d.cross_alpha_host = fp(self.cross_alpha)
        d.mapper_host = fp(self.mapper)
        d.alphas_host = fp(self.alphas)
        d.equalizer_host = fp(self.equalizer)
        d.self_replace_lo, d.self_replace_hi = self.self_range
        d.self_replace_max_tokens = self.self_max_tokens
        d.lb_enabled = 1 if self.lb_alpha is not None else 0
        d.lb_start = self.lb_start
        d.lb_threshold = self.lb_threshold
        d.lb_alpha_host = fp(self.lb_alpha) if self.lb_alpha is not None else None
        d.lb_sub_alpha_host = fp(self.lb_sub_alpha) if self.lb_sub_alpha is not None else None
        d.lb_threshold_sub = self.lb_threshold_sub
        return d


class MasaCtrlTables:
    """MutualSelfAttentionControl (models/masactrl/masactrl.py:14-72) -> pnpi_ctrl_desc kind 2."""

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None):
        """layer_idx / step_idx: explicit lists (any subset of transformer blocks 0..15 / denoising steps); None = the window."""
        self.start_step, self.start_layer = int(start_step), int(start_layer)
        self.layer_mask = 0
        if layer_idx is not None:
            for b in layer_idx:
                if 0 <= int(b) < 32:                      # blocks past the UNet's 16 never match, as in the reference's `in` test
                    self.layer_mask |= 1 << int(b)
            self.layer_mask |= 1 << 31                    # "a list was given" (an empty list switches the control off everywhere)
        self.step_on = None
        if step_idx is not None:
            n = max([int(x) for x in step_idx if int(x) >= 0], default=-1) + 1
            self.step_on = (C.c_ubyte * max(n, 1))()
            for x in step_idx:
                if int(x) >= 0:
                    self.step_on[int(x)] = 1

    def desc(self):
        d = _capi.CtrlDesc()
        d.kind = 2
        d.masa_start_step, d.masa_start_layer = self.start_step, self.start_layer
        d.masa_layer_mask = self.layer_mask
        if self.step_on is not None:
            d.masa_n_steps = len(self.step_on)
            d.masa_step_on_host = C.cast(self.step_on, C.POINTER(C.c_ubyte))
        return d


def _desc_array(ctrls):
    """list[ControllerTables | None] -> (ctypes array of pnpi_ctrl_desc, keep-alive)"""
    if ctrls is None:
        return None
    arr = (_capi.CtrlDesc * len(ctrls))()
    for i, c in enumerate(ctrls):
        if c is None:
            arr[i].kind = 0
        else:
            arr[i] = c.desc()
    return arr


class NativeEngine:
    def __init__(self, cfg: ModelConfig, device=None, max_unet_rows=4, max_vae_images=2, share_weights_with=None):
        """share_weights_with: another NativeEngine on the same device -- this context then borrows its packed weight arena
        (pnpi_create_shared: no second copy of the 1.9 GB) and is ready without load_state_dict; the other engine must outlive it."""
        if not torch.cuda.is_available():
            raise RuntimeError("NativeEngine needs an AMD GPU (gfx950); there is no CPU fallback")
        self.lib = _capi.load_library()
        self.cfg = cfg
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._ccfg = cfg.to_c()
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            if share_weights_with is not None:
                if share_weights_with.device != self.device or share_weights_with.cfg != cfg:
                    raise ValueError("share_weights_with: same device and model configuration required")
                self._weights_owner = share_weights_with          # keep the owner alive as long as this context
                st = self.lib.pnpi_create_shared(C.byref(self.h), share_weights_with.h, C.c_void_p(stream), max_unet_rows, max_vae_images)
            else:
                st = self.lib.pnpi_create(C.byref(self.h), C.byref(self._ccfg), self.device.index or 0, C.c_void_p(stream),
                                          max_unet_rows, max_vae_images)
        if st != 0:
            msg = self.lib.pnpi_last_error(self.h if self.h else (share_weights_with.h if share_weights_with is not None else self.h))
            text = msg.decode() if msg else "?"
            self.close()                 # the half-built context only serves pnpi_last_error: free it
            raise _capi.PnpiError(st, text)
        self.max_unet_rows = max_unet_rows
        self.max_vae_images = max_vae_images
        self.lat_hw = cfg.sample_size
        self.ac = None
        self.final_alpha = None

    # ---- plumbing
    def _call(self, name, *args):
        st = getattr(self.lib, name)(self.h, *args)
        _capi.check(self.lib, self.h, st)

    def close(self):
        if self.h:
            self.lib.pnpi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _f32(self, t):
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    # ---- weights
    def load_state_dict(self, unet_sd=None, vae_sd=None, chunk_bytes=1 << 30, clip_sd=None):
        """diffusers-layout state dicts (CPU or GPU tensors, fp32 or fp16); repacked on the device into the fp16 arena.
        clip_sd: transformers CLIPTextModel state dict (optional; only pnpi_text_encode needs it)."""
        items = []
        for prefix, sd in (("unet.", unet_sd), ("vae.", vae_sd), ("clip.", clip_sd)):
            if sd:
                items += [(prefix + k, v) for k, v in sd.items()]
        batch, keep, size = [], [], 0

        def flush():
            nonlocal batch, keep, size
            if not batch:
                return
            arr = (_capi.NamedTensor * len(batch))(*batch)
            self._call("pnpi_load_weights", arr, len(batch))
            torch.cuda.synchronize(self.device)
            batch, keep, size = [], [], 0

        for name, v in items:
            t = v.detach()
            if t.dtype not in (torch.float32, torch.float16):
                t = t.float()
            t = t.to(self.device).contiguous()
            nt = _capi.NamedTensor()
            nt.name = name.encode()
            nt.data = t.data_ptr()
            nt.dtype = 1 if t.dtype == torch.float16 else 0
            nt.ndim = t.dim()
            for i, s in enumerate(t.shape):
                nt.shape[i] = s
            batch.append(nt)
            keep.append((t, nt.name))
            size += t.numel() * t.element_size()
            if size >= chunk_bytes:
                flush()
        flush()

    def missing_weights(self):
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.pnpi_missing_weights(self.h, buf, len(buf))
        return n, buf.value.decode().split("\n")[:-1]

    def weight_arena(self):
        """The packed weight arena as a uint8 CUDA tensor view (for the one start-up RCCL broadcast)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        self._call("pnpi_weight_arena", C.byref(ptr), C.byref(nbytes))
        return ptr.value, nbytes.value

    def mark_all_loaded(self):
        self._call("pnpi_mark_all_loaded")

    def set_scheduler(self, alphas_cumprod, final_alpha_cumprod):
        ac = np.ascontiguousarray(np.asarray(alphas_cumprod, dtype=np.float32))
        self.ac = ac
        self.final_alpha = float(np.float32(final_alpha_cumprod))
        self._call("pnpi_set_scheduler", ac.ctypes.data_as(C.POINTER(C.c_float)), len(ac), self.final_alpha)

    def counters(self):
        c = _capi.Counters()
        self._call("pnpi_get_counters", C.byref(c))
        return {k: getattr(c, k) for k, _ in c._fields_}

    def reset_counters(self):
        self._call("pnpi_reset_counters")

    def clock_probe(self, iters=100000):
        """effective matrix-pipe clock (GHz) under back-to-back MFMAs on every SIMD, and the probe's duration (ms)"""
        ghz, ms = C.c_float(), C.c_float()
        self._call("pnpi_clock_probe", int(iters), C.byref(ghz), C.byref(ms))
        return float(ghz.value), float(ms.value)

    def profile_begin(self):
        self._call("pnpi_profile_begin")

    def profile_end(self):
        arr = (_capi.KernelStats * len(_capi.KC_NAMES))()
        self._call("pnpi_profile_end", arr)
        return {n: dict(launches=int(arr[i].launches), total_ms=arr[i].total_ms, flops=arr[i].flops, bytes=arr[i].bytes)
                for i, n in enumerate(_capi.KC_NAMES)}

    # ---- level 1
    def text_kv_precompute(self, context):
        """Project the text context to the 16 cross-attention K / V pairs once; unet(..., context=None) then reads them."""
        ctx = self._f32(context)
        self._call("pnpi_text_kv_precompute", _p(ctx), ctx.shape[0])
        self._keep_ctx = ctx

    # ---- level-1 fallback: materialise-and-call-back attention (controllers without a kernel descriptor)
    def attention_sites(self):
        """(tokens, keys) of the 32 attention sites in call order -- sizes the call-back buffer"""
        cfg, S, sites = self.cfg, self.cfg.sample_size, []
        def level(i):
            n = (S >> i) ** 2
            return [(n, n), (n, cfg.ctx_len)]
        for i in range(cfg.n_blocks):
            if cfg.block_has_attn[i]:
                sites += level(i) * cfg.layers_per_block
        sites += level(cfg.n_blocks - 1)
        for i in reversed(range(cfg.n_blocks)):
            if cfg.block_has_attn[i]:
                sites += level(i) * (cfg.layers_per_block + 1)
        return sites

    def set_attention_callback(self, fn, rows=None):
        """fn(attn, is_cross: bool, place: 'down'|'mid'|'up', layer: int) -> None | tensor: called at every attention site of unet()
        with the softmax probabilities as a float32 CUDA tensor [rows*heads, Nq, Nk] (the reference's hooked forward,
        models/p2p/attention_control.py:40-44); it may modify the tensor in place or return a replacement.  fn = None removes it."""
        if fn is None:
            self._call("pnpi_set_attention_callback", None, None, None, 0)
            self._cb_keep = None
            return
        rows = rows or self.max_unet_rows
        heads = self.cfg.heads
        need = max(rows * heads * nq * nk * 4 + nq * ((nk + 7) // 8 * 8) * 2 + 512 for nq, nk in self.attention_sites())
        buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        base = buf.data_ptr()
        places = ("down", "mid", "up")
        err = []

        def trampoline(user, attn_ptr, B, H, Nq, Nk, is_cross, place, layer):
            try:
                n = B * H * Nq * Nk
                off = attn_ptr - base
                view = buf[off:off + 4 * n].view(torch.float32).view(B * H, Nq, Nk)
                with torch.cuda.device(self.device):
                    new = fn(view, bool(is_cross), places[place], layer)
                    if new is not None and new is not view:
                        view.copy_(new.reshape(view.shape))
                    torch.cuda.current_stream().synchronize()
                return 0
            except Exception as e:       # never let an exception cross the C boundary
                err.append(e)
                return 1

        cb = _capi.ATTN_CALLBACK(trampoline)
        self._call("pnpi_set_attention_callback", cb, None, C.c_void_p(base), need)
        self._cb_keep = (cb, buf, err)

    def unet(self, latents, t, context, rows_per_image=1, ctrls=None, cur_step=0):
        lat, ctx = self._f32(latents), (self._f32(context) if context is not None else None)
        rows = lat.shape[0]
        out = torch.empty_like(lat)
        arr = _desc_array(ctrls)
        try:
            self._call("pnpi_unet_forward", _p(lat), rows, rows_per_image, int(t), _p(ctx), arr, int(cur_step), _p(out))
        except _capi.PnpiError:
            keep = getattr(self, "_cb_keep", None)
            if keep and keep[2]:
                raise keep[2].pop()          # the exception raised inside the attention callback
            raise
        self._keep = (lat, ctx, arr, ctrls)
        return out

    def text_encode(self, input_ids):
        """CLIPTextModel(input_ids)[0] on the device: int ids [n, 77] -> fp32 [n, 77, cross_dim]"""
        ids = torch.as_tensor(input_ids).to(self.device, dtype=torch.int32).contiguous()
        n, T = ids.shape
        if T != self.cfg.ctx_len:
            raise ValueError("input_ids must have %d positions" % self.cfg.ctx_len)
        out = torch.empty(n, T, self.cfg.cross_dim, device=self.device)
        self._call("pnpi_text_encode", _p(ids), n, _p(out))
        self._keep = ids
        return out

    def local_blend(self, latents, step_index):
        lat = self._f32(latents)
        self._call("pnpi_local_blend", _p(lat), lat.shape[0] // 2, int(step_index))
        return lat

    def vae_encode(self, x):
        x = self._f32(x)
        n, _, H, W = x.shape
        f = self.cfg.vae_scale
        out = torch.empty(n, self.cfg.vae_latent_channels, H // f, W // f, device=self.device)
        self._call("pnpi_vae_encode", _p(x), n, H, W, _p(out))
        self._keep = x
        return out

    def vae_decode(self, z):
        z = self._f32(z)
        n, _, h, w = z.shape
        f = self.cfg.vae_scale
        out = torch.empty(n, 3, h * f, w * f, device=self.device)
        self._call("pnpi_vae_decode", _p(z), n, h, w, _p(out))
        self._keep = z
        return out

    def image2latent(self, img_u8):
        """uint8 [n,H,W,3] (or [H,W,3]) -> 0.18215 * posterior mean, fp32 [n,4,H/8,W/8]."""
        img = torch.as_tensor(img_u8)
        if img.dim() == 3:
            img = img[None]
        img = img.to(self.device).contiguous()
        n, H, W, _ = img.shape
        f = self.cfg.vae_scale
        out = torch.empty(n, self.cfg.vae_latent_channels, H // f, W // f, device=self.device)
        self._call("pnpi_image2latent", _p(img), n, H, W, _p(out))
        self._keep = img
        return out

    def latent2image(self, z):
        z = self._f32(z)
        n, _, h, w = z.shape
        f = self.cfg.vae_scale
        out = torch.empty(n, h * f, w * f, 3, dtype=torch.uint8, device=self.device)
        self._call("pnpi_latent2image", _p(z), n, h, w, _p(out))
        self._keep = z
        return out

    def ddim_next_step(self, eps, t, ratio, sample):
        e, s = self._f32(eps), self._f32(sample)
        out = torch.empty_like(s)
        self._call("pnpi_ddim_next_step", _p(e), int(t), int(ratio), _p(s), s.numel(), _p(out))
        self._keep = (e, s)
        return out

    def ddim_prev_step(self, eps, t, ratio, sample):
        e, s = self._f32(eps), self._f32(sample)
        out = torch.empty_like(s)
        self._call("pnpi_ddim_prev_step", _p(e), int(t), int(ratio), _p(s), s.numel(), _p(out))
        self._keep = (e, s)
        return out

    def cfg_ddim_prev(self, eps, x, t, ratio, guidance_scale, noise_loss=None, offset_rows=0):
        """One image's CFG combine + DDIM step (+ the direct-inversion offset on the first offset_rows rows): eps [2R, 4, h, w]
        (R unconditional rows, then R conditional), x [R, 4, h, w] -> [R, 4, h, w]   (p2p_guidance_forward.py:108-114)"""
        e, xs = self._f32(eps), self._f32(x)
        R = xs.shape[0]
        assert e.shape[0] == 2 * R
        nl = self._f32(noise_loss) if noise_loss is not None else None
        out = torch.empty_like(xs)
        self._call("pnpi_cfg_ddim_prev", _p(e), _p(xs), 1, R, xs[0].numel(), float(guidance_scale), int(t), int(ratio), _p(nl),
                   int(offset_rows) if nl is not None else 0, None, 1.0, None, _p(out), None, 0, None)
        self._keep = (e, xs, nl)
        return out

    def ddim_prev_step_recon(self, eps, t, ratio, sample, ref_image, recon_lr, recon_mask=None):
        """DDIMSchedulerDev.step with ref_image / recon_lr / recon_mask (scheduler_dev.py:68-76) -> (prev_sample, pred_original_sample)"""
        e, s = self._f32(eps), self._f32(sample)
        ref = self._f32(ref_image.to(self.device).expand_as(s))
        mask = None if recon_mask is None else self._f32(recon_mask.to(self.device).expand_as(s).float())
        out, x0 = torch.empty_like(s), torch.empty_like(s)
        self._call("pnpi_ddim_prev_step_recon", _p(e), int(t), int(ratio), _p(s), s.numel(), _p(ref), float(recon_lr),
                   _p(mask) if mask is not None else None, _p(out), _p(x0))
        self._keep = (e, s, ref, mask)
        return out, x0

    # ---- level 2
    def _ts(self, timesteps):
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int32))
        return ts, ts.ctypes.data_as(C.POINTER(C.c_int))

    def _recon_desc(self, recon, nimg, xT, nsteps):
        """recon dict -> (pnpi_recon_desc, tensors to keep alive).  ref_image: [nimg,4,h,w] encoded source latent or None; x_stars: None
        or the inversion trajectory [nsteps+1, 1 | nimg, 4, h, w] (inversion guidance, proximal_guidance_forward.py:73-75)."""
        if recon is None:
            return None, None
        ref = inv = None
        if recon.get("ref_image") is not None:
            ref = self._f32(recon["ref_image"]).reshape(nimg, *xT.shape[1:]).contiguous()
        if recon.get("x_stars") is not None:
            xs = recon["x_stars"]
            xs = torch.stack([x for x in xs]) if isinstance(xs, (list, tuple)) else xs
            inv = self._f32(xs).reshape(nsteps + 1, -1, *xT.shape[1:])
            inv = inv.expand(nsteps + 1, nimg, *xT.shape[1:]).contiguous()
        rd = _capi.ReconDesc.make(ref.data_ptr() if ref is not None else None, recon["recon_lr"], recon["recon_t"],
                                  int(recon.get("dilate_mask") or 0), inv.data_ptr() if inv is not None else None)
        return rd, (ref, inv)

    def edit_loop_uncond_steps(self, x_T, context4, uncond_steps, ctrls, timesteps, guidance_scale, first_only=False, prox=None, quantile=0.7,
                               recon=None):
        """edit_loop with per-step unconditional embeddings [steps, nimg, 77, D] (null-text inversion); recon as in edit_loop."""
        xT, ctx, us = self._f32(x_T), self._f32(context4), self._f32(uncond_steps)
        nimg = xT.shape[0]
        out = torch.empty(nimg, 2, *xT.shape[1:], device=self.device)
        ts, tsp = self._ts(timesteps)
        arr = _desc_array(ctrls)
        mode = {None: 0, "l0": 1, "l1": 2}[prox]
        rd, ref = self._recon_desc(recon, nimg, xT, len(timesteps))
        self._call("pnpi_edit_loop_uncond_steps_recon", _p(xT), nimg, _p(ctx), arr, len(timesteps), tsp, float(guidance_scale), mode, float(quantile),
                   _p(us), int(bool(first_only)), C.byref(rd) if rd is not None else None, _p(out))
        self._keep = (xT, ctx, us, ts, arr, ctrls, ref, rd)
        return out

    def unet_context_grad(self, latents, t, context, d_eps):
        """eps and d loss / d context for ONE row (null-text path groundwork): latents [1,4,h,w], context [1,77,D], d_eps like eps."""
        lat, ctx, de = self._f32(latents), self._f32(context), self._f32(d_eps)
        eps, dctx = torch.empty_like(lat), torch.empty_like(ctx)
        self._call("pnpi_unet_context_grad", _p(lat), int(t), _p(ctx), _p(de), _p(eps), _p(dctx))
        return eps, dctx

    def null_text_optimize(self, ddim_latents, ctx_uncond, ctx_cond, timesteps, guidance_scale, num_inner_steps=10, epsilon=1e-5,
                           return_losses=False):
        """NullInversion.null_optimization for one image -> ([steps, 1, 77, D] embeddings, [steps] Adam iterations run[, per-step lists
        of every iteration's loss])."""
        lat = self._f32(ddim_latents)
        n = lat.shape[0] - 1
        cu, cc = self._f32(ctx_uncond), self._f32(ctx_cond)
        out = torch.empty(n, 1, cu.shape[-2], cu.shape[-1], device=self.device, dtype=torch.float32)
        its = (C.c_int * n)()
        losses = (C.c_float * max(1, n * int(num_inner_steps)))()
        ts_keep, ts_ptr = self._ts(timesteps)
        self._call("pnpi_null_text_optimize", _p(lat), _p(cu), _p(cc), n, ts_ptr, float(guidance_scale), int(num_inner_steps),
                   float(epsilon), _p(out), its, losses)
        del ts_keep
        if return_losses:
            return out, list(its), [[losses[i * num_inner_steps + j] for j in range(its[i])] for i in range(n)]
        return out, list(its)

    def null_latent_calculate(self, ddim_latents, context4, timesteps, guidance_scale, num_inner_steps=10, epsilon=1e-5, return_losses=False):
        """DirectInversion.null_latent_calculate for one prompt pair -> [steps, 2, 4, h, w] latent offsets (context4 rows:
        unc_src, unc_tgt, cond_src, cond_tgt)."""
        lat = self._f32(ddim_latents)
        n = lat.shape[0] - 1
        c4 = self._f32(context4)
        assert c4.shape[0] == 4, "null_latent_calculate handles one (source, target) prompt pair"
        out = torch.empty(n, 2, *lat.shape[-3:], device=self.device, dtype=torch.float32)
        its = (C.c_int * n)()
        losses = (C.c_float * max(1, n * int(num_inner_steps)))()
        ts_keep, ts_ptr = self._ts(timesteps)
        self._call("pnpi_null_latent_calculate", _p(lat), _p(c4), n, ts_ptr, float(guidance_scale), int(num_inner_steps), float(epsilon),
                   _p(out), its, losses)
        del ts_keep
        if return_losses:
            return out, list(its), [[losses[i * num_inner_steps + j] for j in range(its[i])] for i in range(n)]
        return out, list(its)

    def ddim_invert(self, z0, ctx_cond, timesteps):
        z0, ctx = self._f32(z0), self._f32(ctx_cond)
        n = len(timesteps)
        nimg = z0.shape[0]
        out = torch.empty(n + 1, *z0.shape, device=self.device)
        ts, tsp = self._ts(timesteps)
        self._call("pnpi_ddim_invert", _p(z0), nimg, _p(ctx), n, tsp, _p(out))
        self._keep = (z0, ctx, ts)
        return out

    @staticmethod
    def _scales(offset_scale, n):
        """None | float | sequence[n] -> (keep-alive array, float* or None): per-step factor on the direct-inversion offset"""
        if offset_scale is None:
            return None, None
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(offset_scale, dtype=np.float32), (n,)))
        return sc, sc.ctypes.data_as(C.POINTER(C.c_float))

    def ddim_invert_cfg(self, z0, ctx_uncond, ctx_cond, timesteps, guidance_scale):
        """DDIM inversion under classifier-free guidance (inversion.py:334-347) -> [steps+1, nimg, 4, h, w]"""
        z0, cu, cc = self._f32(z0), self._f32(ctx_uncond), self._f32(ctx_cond)
        n, nimg = len(timesteps), z0.shape[0]
        out = torch.empty(n + 1, *z0.shape, device=self.device)
        ts, tsp = self._ts(timesteps)
        self._call("pnpi_ddim_invert_cfg", _p(z0), nimg, _p(cu), _p(cc), float(guidance_scale), n, tsp, _p(out))
        self._keep = (z0, cu, cc, ts)
        return out

    def offset_calculate(self, ddim_latents, context4, timesteps, guidance_scale, offset_scale=None):
        lat, ctx = self._f32(ddim_latents), self._f32(context4)
        n = len(timesteps)
        nimg = lat.shape[1]
        out = torch.empty(n, nimg, 2, *lat.shape[2:], device=self.device)
        ts, tsp = self._ts(timesteps)
        sc, scp = self._scales(offset_scale, n)
        self._call("pnpi_offset_calculate", _p(lat), nimg, _p(ctx), n, tsp, float(guidance_scale), scp, _p(out))
        self._keep = (lat, ctx, ts, sc)
        return out

    def direct_edit(self, ddim_latents, context4, ctrls_per_pass, timesteps, guidance_scale, offset_rows=1, offset_scale=None):
        """offset_calculate + len(ctrls_per_pass) guidance-forward passes in lock step (pnpi_direct_edit).
        ctrls_per_pass: list (passes) of None | list[ControllerTables | None] (images).  -> (noise_loss, latents[npass])"""
        lat, ctx = self._f32(ddim_latents), self._f32(context4)
        n = len(timesteps)
        nimg = lat.shape[1]
        npass = len(ctrls_per_pass)
        flat = []
        for cp in ctrls_per_pass:
            flat += list(cp) if cp is not None else [None] * nimg
        assert len(flat) == npass * nimg
        arr = _desc_array(flat)
        nl = torch.empty(n, nimg, 2, *lat.shape[2:], device=self.device)
        out = torch.empty(npass, nimg, 2, *lat.shape[2:], device=self.device)
        ts, tsp = self._ts(timesteps)
        sc, scp = self._scales(offset_scale, n)
        self._call("pnpi_direct_edit", _p(lat), nimg, _p(ctx), npass, arr, int(offset_rows), n, tsp, float(guidance_scale), scp, _p(nl), _p(out))
        self._keep = (lat, ctx, ts, arr, flat, sc)
        return nl, out

    def direct_edit_pruned(self, ddim_latents, context4, ctrls, timesteps, guidance_scale):
        """The pruned-equivalent schedule (SURVEY Note D): 3 rows per image and step, source latent assigned from the trajectory.
        ctrls: None | list[ControllerTables | None] per image.  -> latents [nimg, 2, 4, h, w] = (x*_0, edited)"""
        lat, ctx = self._f32(ddim_latents), self._f32(context4)
        n, nimg = len(timesteps), lat.shape[1]
        arr = _desc_array(ctrls)
        out = torch.empty(nimg, 2, *lat.shape[2:], device=self.device)
        ts, tsp = self._ts(timesteps)
        self._call("pnpi_direct_edit_pruned", _p(lat), nimg, _p(ctx), arr, n, tsp, float(guidance_scale), _p(out))
        self._keep = (lat, ctx, ts, arr, ctrls)
        return out

    def edit_loop(self, x_T, context4, noise_loss, ctrls, timesteps, guidance_scale, offset_rows=1, prox=None, quantile=0.7,
                  recon=None):
        """recon: None | dict(ref_image=[nimg,4,h,w] encoded source latent, recon_lr, recon_t, dilate_mask): reconstruction guidance
        (proximal_guidance_forward.py:48-51 + scheduler_dev.py:68-76), applied with prox 'l0' / 'l1' only."""
        xT, ctx = self._f32(x_T), self._f32(context4)
        nl = self._f32(noise_loss) if noise_loss is not None else None
        n = len(timesteps)
        nimg = xT.shape[0]
        out = torch.empty(nimg, 2, *xT.shape[1:], device=self.device)
        ts, tsp = self._ts(timesteps)
        arr = _desc_array(ctrls)
        mode = {None: 0, "l0": 1, "l1": 2}[prox]
        rd, ref = self._recon_desc(recon, nimg, xT, n)
        self._call("pnpi_edit_loop", _p(xT), nimg, _p(ctx), _p(nl), int(offset_rows), arr, n, tsp, float(guidance_scale), mode,
                   float(quantile), C.byref(rd) if rd is not None else None, _p(out))
        self._keep = (xT, ctx, nl, ts, arr, ctrls, ref, rd)
        return out

"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the direct-inversion + Prompt-to-Prompt loop: scheduler tables, DDIM moves,
the three DI lines, the attention controllers and LocalBlend, restated in plain PyTorch/numpy from

  models/p2p/inversion.py:245-400          (DirectInversion: next_step, prev_step, ddim_loop, offset_calculate)
  models/p2p/inversion.py:196-234          (NullInversion.null_optimization / invert: the oracle of the not-yet-built null-text path)
  models/p2p/p2p_guidance_forward.py:103-173 (direct_inversion_p2p_guidance_{diffusion_step,forward})
  models/p2p/scheduler_dev.py:38-95        (DDIMSchedulerDev.step, eta = 0)
  models/p2p/attention_control.py:95-363   (LocalBlend, AttentionStore, AttentionControlEdit, Replace/Refine/Reweight)
  utils/utils.py:58-80                     (latent2image, image2latent)

Scalars: the reference evaluates `alpha ** 0.5` on 0-dim fp32 tensors; torch's CPU pow is 1 ulp off a correctly rounded
sqrt on some hosts, so this oracle (like the HIP kernels) pins the IEEE answer: correctly rounded fp32 sqrt.
Pinned against the reference's own code by oracle/make_golden.py -> tests/golden/ (see tests/test_oracle_golden.py)."""
import numpy as np
import torch
import torch.nn.functional as F

MAX_NUM_WORDS = 77


# ----------------------------------------------------------------------------------------------------- scheduler tables
def alphas_cumprod(n_train=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler(beta_schedule="scaled_linear") tables as built in models/p2p_editor.py:18-22 (diffusers 0.10.0)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def make_timesteps(num_inference_steps, n_train=1000):
    ratio = n_train // num_inference_steps
    return (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)


def _scalars(a_from, a_to, dtype):
    if dtype == torch.float32:
        f = np.float32
        af, at = f(a_from), f(a_to)
        return float(np.sqrt(af)), float(np.sqrt(f(1) - af)), float(np.sqrt(at)), float(np.sqrt(f(1) - at))
    af, at = float(a_from), float(a_to)
    return af ** 0.5, (1 - af) ** 0.5, at ** 0.5, (1 - at) ** 0.5


def ddim_move(x, eps, a_from, a_to):
    """x_to = sqrt(a_to) * (x - sqrt(1-a_from) eps) / sqrt(a_from) + sqrt(1-a_to) eps   (inversion.py:251-253, 266-269)."""
    sa_f, sb_f, sa_t, sb_t = _scalars(a_from, a_to, x.dtype)
    x0 = (x - sb_f * eps) / sa_f
    return sa_t * x0 + sb_t * eps


def next_alphas(ac, final, t, ratio):
    tp = min(t - ratio, 999)
    return (ac[tp] if tp >= 0 else final), ac[t]


def prev_alphas(ac, final, t, ratio):
    tp = t - ratio
    return ac[t], (ac[tp] if tp >= 0 else final)


# ----------------------------------------------------------------------------------------------------- controllers
class StoreController:
    """AttentionStore (attention_control.py:214-248): keeps the conditional half of every <= 32^2-token map, summed over steps."""

    def __init__(self, num_att_layers):
        self.num_att_layers = num_att_layers
        self.cur_step = 0
        self.cur_att_layer = 0
        self.step_store = self._empty()
        self.attention_store = {}

    @staticmethod
    def _empty():
        return {"down_cross": [], "mid_cross": [], "up_cross": [], "down_self": [], "mid_self": [], "up_self": []}

    def edit(self, attn, is_cross, place):
        return attn

    def __call__(self, attn, is_cross, place):
        h = attn.shape[0]
        cond = self.edit(attn[h // 2:], is_cross, place)
        if cond.shape[1] <= 32 ** 2:
            self.step_store["%s_%s" % (place, "cross" if is_cross else "self")].append(cond)
        attn = torch.cat([attn[: h // 2], cond], dim=0)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            if not self.attention_store:
                self.attention_store = self.step_store
            else:
                for k in self.attention_store:
                    for i in range(len(self.attention_store[k])):
                        self.attention_store[k][i] = self.attention_store[k][i] + self.step_store[k][i]
            self.step_store = self._empty()
        return attn

    def step_callback(self, x_t):
        return x_t


class EditController(StoreController):
    """AttentionControlEdit.forward (:269-282) with AttentionReplace (:303-304), AttentionRefine (:319-323),
    AttentionReweight (:340-345) and LocalBlend (:97-121).  `tables` are the host tables of make_controller (:366-405):
      cross_alpha [steps+1, 77]; kind "replace" (mapper [77,77]) or "refine" (mapper [77] int64, alphas [77]) or "none";
      equalizer [77] or None; self_range (lo, hi); lb = dict(alpha_layers [2,77], start, th[, sub_alpha_layers [2,77], th_sub]) or None."""

    def __init__(self, num_att_layers, tables, batch_size=2):
        super().__init__(num_att_layers)
        self.t = tables
        self.batch_size = batch_size
        self.lb_counter = 0

    def _replace_cross(self, base, repl):
        t = self.t
        if t["kind"] == "replace":
            out = torch.einsum("hpw,bwn->bhpn", base, t["mapper"].to(base.dtype)[None])
        elif t["kind"] == "refine":
            gathered = base[:, :, t["mapper"][None]].permute(2, 0, 1, 3)
            al = t["alphas"].to(base.dtype).reshape(1, 1, 1, -1)
            out = gathered * al + repl * (1 - al)
        else:
            out = base[None]
        if t.get("equalizer") is not None:
            # AttentionReweight: attn_base[None] * equalizer[:, None, None, :]; the blend keeps the extra leading dim and the
            # assignment broadcasts it back
            out = out.reshape(-1, *out.shape[-3:]) * t["equalizer"].to(base.dtype).reshape(1, 1, 1, -1)
        return out

    def edit(self, attn, is_cross, place):
        lo, hi = self.t["self_range"]
        if is_cross or (lo <= self.cur_step < hi):
            h = attn.shape[0] // self.batch_size
            a = attn.reshape(self.batch_size, h, *attn.shape[1:])
            base, repl = a[0], a[1:]
            if is_cross:
                aw = self.t["cross_alpha"][self.cur_step].to(attn.dtype).reshape(1, 1, 1, -1)
                new = self._replace_cross(base, repl) * aw + (1 - aw) * repl
            else:
                new = base[None].expand_as(repl) if repl.shape[2] <= 32 ** 2 else repl
            attn = torch.cat([base[None], new], dim=0).reshape(self.batch_size * h, *attn.shape[1:])
        return attn

    @staticmethod
    def _lb_get_mask(maps, alpha, use_pool, th, size):
        # LocalBlend.get_mask (:97-106)
        m = (maps * alpha).sum(-1).mean(1)
        if use_pool:
            m = F.max_pool2d(m, (3, 3), (1, 1), padding=(1, 1))
        m = F.interpolate(m, size=size)
        m = m / m.max(2, keepdims=True)[0].max(3, keepdims=True)[0]
        m = m.gt(th)
        return m[:1] + m

    def local_blend_mask(self, x_t):
        lb = self.t["lb"]
        maps = self.attention_store["down_cross"][2:4] + self.attention_store["up_cross"][:3]
        al = lb["alpha_layers"].to(x_t.dtype).reshape(2, 1, 1, 1, 1, MAX_NUM_WORDS)
        maps = [m.reshape(2, -1, 1, 16, 16, MAX_NUM_WORDS) for m in maps]
        maps = torch.cat(maps, dim=1)
        mask = self._lb_get_mask(maps, al, True, lb["th"], x_t.shape[2:])
        if lb.get("sub_alpha_layers") is not None:
            # substruct_words (:114-116): mask * ~get_mask(maps, substruct_layers, use_pool=False), thresholded at th[1]
            sub = lb["sub_alpha_layers"].to(x_t.dtype).reshape(2, 1, 1, 1, 1, MAX_NUM_WORDS)
            mask = mask * ~self._lb_get_mask(maps, sub, False, lb.get("th_sub", 0.3), x_t.shape[2:])
        return mask

    def step_callback(self, x_t):
        lb = self.t.get("lb")
        if lb is not None:
            self.lb_counter += 1
            if self.lb_counter > lb["start"]:
                mask = self.local_blend_mask(x_t).to(x_t.dtype)
                x_t = x_t[:1] + mask * (x_t - x_t[:1])
        return x_t


class MasaCtrlEditor:
    """AttentionBase.__call__ + MutualSelfAttentionControl.forward (models/masactrl/masactrl_utils.py:14-41,
    masactrl.py:41-69): at steps >= start_step and transformer blocks >= start_layer the self-attention of every row of a CFG
    half uses K, V of the half's first row."""

    def __init__(self, start_step=4, start_layer=10, num_att_layers=32):
        self.start_step, self.start_layer, self.num_att_layers = start_step, start_layer, num_att_layers
        self.cur_step = 0
        self.cur_att_layer = 0

    @staticmethod
    def _merge(out, heads):          # '(b h) n d -> b n (h d)'
        bh, n, d = out.shape
        return out.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, heads * d)

    def _attn_batch(self, q, k, v, heads, scale):   # masactrl.py:41-55: all rows' queries against one row's keys / values
        b = q.shape[0] // heads
        n, d = q.shape[1], q.shape[2]
        qq = q.reshape(b, heads, n, d).permute(1, 0, 2, 3).reshape(heads, b * n, d)
        sim = torch.einsum("hid,hjd->hij", qq, k) * scale
        out = torch.einsum("hij,hjd->hid", sim.softmax(-1), v)                      # [h, b*n, d]
        return out.reshape(heads, b, n, d).permute(1, 2, 0, 3).reshape(b, n, heads * d)

    def qkv_editor(self, q, k, v, sim, attn, is_cross, place, heads, scale):
        if is_cross or self.cur_step < self.start_step or self.cur_att_layer // 2 < self.start_layer:
            out = self._merge(torch.einsum("bij,bjd->bid", attn, v), heads)
        else:
            qu, qc = q.chunk(2)
            ku, kc = k.chunk(2)
            vu, vc = v.chunk(2)
            out = torch.cat([self._attn_batch(qu, ku[:heads], vu[:heads], heads, scale),
                             self._attn_batch(qc, kc[:heads], vc[:heads], heads, scale)])
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
        return out

    def step_callback(self, x_t):
        return x_t


# ----------------------------------------------------------------------------------------------------- loops
def ddim_loop(unet_fn, z0, ctx_cond, timesteps, ac, final):
    """DirectInversion.ddim_loop (inversion.py:308-319): B = 1, conditional source embedding only, t ascending."""
    n = len(timesteps)
    ratio = len(ac) // n
    lat = z0.clone()
    all_lat = [z0]
    for i in range(n):
        t = int(timesteps[n - i - 1])
        eps = unet_fn(lat, t, ctx_cond, None)
        a_from, a_to = next_alphas(ac, final, t, ratio)
        lat = ddim_move(lat, eps, float(a_from), float(a_to))
        all_lat.append(lat)
    return all_lat


def ddim_loop_cfg(unet_fn, z0, ctx_uncond, ctx_cond, timesteps, ac, final, guidance_scale):
    """DirectInversion.ddim_with_guidance_scale_loop (inversion.py:334-347): two B = 1 UNet calls per step, CFG, next_step."""
    n = len(timesteps)
    ratio = len(ac) // n
    lat = z0.clone()
    all_lat = [z0]
    for i in range(n):
        t = int(timesteps[n - i - 1])
        eu = unet_fn(lat, t, ctx_uncond, None)
        ec = unet_fn(lat, t, ctx_cond, None)
        e = eu + guidance_scale * (ec - eu)
        a_from, a_to = next_alphas(ac, final, t, ratio)
        lat = ddim_move(lat, e, float(a_from), float(a_to))
        all_lat.append(lat)
    return all_lat


def offset_calculate(unet_fn, ddim_latents, context4, timesteps, ac, final, guidance_scale, offset_scale=None):
    """DirectInversion.offset_calculate (inversion.py:375-391).  context4 rows = [unc_src, unc_tgt, cond_src, cond_tgt].
    offset_scale: None | float (offset_calculate_not_full, :478-493: loss *= scale) | per-step list (offset_calculate_skip_step,
    :502-519: loss = 0 on the steps with i % skip_step != 0)."""
    n = len(timesteps)
    ratio = len(ac) // n
    nrow = context4.shape[0] // 2
    cur = torch.cat([ddim_latents[-1]] * nrow)
    losses = []
    for i in range(n):
        prev_target = torch.cat([ddim_latents[len(ddim_latents) - i - 2]] * cur.shape[0])
        t = int(timesteps[i])
        eps = unet_fn(torch.cat([cur] * 2), t, context4, None)
        eu, ec = eps.chunk(2)
        e = eu + guidance_scale * (ec - eu)
        a_t, a_p = prev_alphas(ac, final, t, ratio)
        prev_rec = ddim_move(cur, e, float(a_t), float(a_p))
        loss = prev_target - prev_rec
        if offset_scale is not None:
            sc = offset_scale[i] if hasattr(offset_scale, "__len__") else offset_scale
            loss = loss * sc if sc != 0 else torch.zeros_like(loss)
        losses.append(loss)
        cur = prev_rec + loss
    return losses


def guidance_forward(unet_fn, x_T, context4, noise_loss_list, controller, timesteps, ac, final, guidance_scale, offset_rows=1,
                     collect=None, prox=None, quantile=0.7, recon=None, uncond_list=None, uncond_first_only=False):
    """direct_inversion_p2p_guidance_forward (p2p_guidance_forward.py:135-173) + ..._diffusion_step (:103-116).
    prox 'l0' / 'l1': the proximal step of proximal_guidance_diffusion_step (proximal_guidance_forward.py:39-64), no inversion
    guidance (what the reference's editors reach).  recon = dict(ref_image, recon_lr, recon_t, dilate_mask): reconstruction
    guidance (:48-51,60-72 + DDIMSchedulerDev.step's ref_image / recon_mask branch, scheduler_dev.py:68-76).
    uncond_list: per-step [1, 77, D] unconditional embeddings (null-text inversion).  p2p_guidance_forward (p2p_guidance_forward.py:
    21-62) uses the step's embedding for EVERY unconditional row (`uncond_embeddings[i].expand(*text_embeddings.shape)`);
    uncond_first_only = the single-branch variant (:65-100: `cat([uncond_embeddings[i], uncond_embeddings_[1:]])`)."""
    n = len(timesteps)
    ratio = len(ac) // n
    nrow = context4.shape[0] // 2
    lat = x_T.expand(nrow, *x_T.shape[1:]).clone()
    for i in range(n):
        t = int(timesteps[i])
        if uncond_list is None:
            ctx_i = context4
        elif uncond_first_only:
            ctx_i = torch.cat([uncond_list[i], context4[1:]])
        else:
            ctx_i = torch.cat([uncond_list[i].expand(nrow, *uncond_list[i].shape[1:]), context4[nrow:]])
        eps = unet_fn(torch.cat([lat] * 2), t, ctx_i, controller)
        eu, ec = eps.chunk(2)
        d = ec - eu
        if prox is not None:
            thr = d.abs().quantile(quantile) if quantile > 0 else -quantile
            d = d - d.clamp(-thr, thr)
            if prox == "l1":
                d = torch.where(d > 0, d - thr, d)
                d = torch.where(d < 0, d + thr, d)
        e = eu + guidance_scale * d
        a_t, a_p = prev_alphas(ac, final, t, ratio)
        rt = recon["recon_t"] if recon is not None else 0
        if prox is not None and recon is not None and recon["recon_lr"] > 0 and ((rt > 0 and t < rt) or (rt < 0 and t > -rt)):
            mask_edit = (d.abs() > thr).float()
            if recon.get("dilate_mask", 0) > 0:
                r_ = int(recon["dilate_mask"])
                mask_edit = F.max_pool2d(mask_edit, 2 * r_ + 1, 1, r_)
            recon_mask = 1 - mask_edit
            sa_f, sb_f, sa_t, sb_t = _scalars(float(a_t), float(a_p), lat.dtype)
            x0 = (lat - sb_f * e) / sa_f
            x0 = x0 - recon["recon_lr"] * (x0 - recon["ref_image"].expand_as(x0)) * recon_mask
            lat = sa_t * x0 + sb_t * e
        else:
            lat = ddim_move(lat, e, float(a_t), float(a_p))
        if noise_loss_list is not None:
            lat = torch.cat((lat[:offset_rows] + noise_loss_list[i][:offset_rows], lat[offset_rows:]))
        if controller is not None:
            lat = controller.step_callback(lat)
        if collect is not None:
            collect.append(lat.clone())
    return lat


def null_optimization(unet_fn, ddim_latents, ctx_uncond, ctx_cond, timesteps, ac, final, guidance_scale, num_inner_steps=10,
                      epsilon=1e-5, trace=None, total_steps=None):
    """NullInversion.null_optimization (inversion.py:196-225): per DDIM step (t descending) up to `num_inner_steps` Adam iterations on
    the 77 x D unconditional embedding so that one CFG prev_step from latent_cur lands on the stored inversion latent; then the step
    is taken with the optimised embedding.  unet_fn must be differentiable w.r.t. its context argument (the oracle's UNet is plain
    torch).  Adam is written out (torch.optim.Adam defaults: betas 0.9 / 0.999, eps 1e-8, no weight decay; a NEW optimiser per
    DDIM step, lr = 1e-2 (1 - i / 100)) so that a device implementation has the formula to match.  Returns the list of [1, 77, D]
    embeddings; trace (a list) receives (step, inner iterations run, last loss, [every iteration's loss]).  total_steps: the schedule length when `timesteps` is
    only its first part (tests)."""
    n = len(timesteps)
    ratio = len(ac) // (total_steps or n)
    uncond = ctx_uncond.clone()
    out = []
    latent_cur = ddim_latents[-1]
    for i in range(n):
        uncond = uncond.clone().detach()
        t = int(timesteps[i])
        a_t, a_p = prev_alphas(ac, final, t, ratio)
        its, loss_item, all_losses = 0, None, []
        if num_inner_steps != 0:
            lr = 1e-2 * (1.0 - i / 100.0)
            m, v = torch.zeros_like(uncond), torch.zeros_like(uncond)
            latent_prev = ddim_latents[len(ddim_latents) - i - 2]
            with torch.no_grad():
                eps_cond = unet_fn(latent_cur, t, ctx_cond, None)
            for j in range(num_inner_steps):
                u = uncond.clone().detach().requires_grad_(True)
                with torch.enable_grad():
                    eps_unc = unet_fn(latent_cur, t, u, None)
                    eps = eps_unc + guidance_scale * (eps_cond - eps_unc)
                    rec = ddim_move(latent_cur, eps, float(a_t), float(a_p))
                    loss = F.mse_loss(rec, latent_prev)
                g, = torch.autograd.grad(loss, u)
                k = j + 1
                m = 0.9 * m + 0.1 * g
                v = 0.999 * v + 0.001 * g * g
                denom = (v.sqrt() / (1 - 0.999 ** k) ** 0.5) + 1e-8
                uncond = (uncond - (lr / (1 - 0.9 ** k)) * (m / denom)).detach()
                its, loss_item = k, float(loss.detach())
                all_losses.append(loss_item)
                if loss_item < epsilon + i * 2e-5:
                    break
        if trace is not None:
            trace.append((i, its, loss_item, all_losses))
        out.append(uncond[:1].detach())
        with torch.no_grad():
            eu = unet_fn(latent_cur, t, uncond, None)
            ec = unet_fn(latent_cur, t, ctx_cond, None)
            latent_cur = ddim_move(latent_cur, eu + guidance_scale * (ec - eu), float(a_t), float(a_p))
    return out


def _adam_update(p, g, m, v, k, lr):
    """torch.optim.Adam defaults, step k >= 1 (see null_optimization)."""
    m = 0.9 * m + 0.1 * g
    v = 0.999 * v + 0.001 * g * g
    denom = (v.sqrt() / (1 - 0.999 ** k) ** 0.5) + 1e-8
    return (p - (lr / (1 - 0.9 ** k)) * (m / denom)).detach(), m, v


def null_latent_calculate(unet_fn, ddim_latents, context4, timesteps, ac, final, guidance_scale, num_inner_steps=10, epsilon=1e-5,
                          trace=None):
    """DirectInversion.null_latent_calculate (inversion.py:418-460; "ablation_null-latent-inversion+p2p"): per step the unconditional
    embeddings of BOTH prompts are optimised like null-text inversion (one B = 2 CFG forward per Adam iteration, the loss on the SOURCE
    row only, so the target row's embedding receives a zero gradient and stays put), then the step's effect is converted into a latent
    offset: noise_loss[i] = prev_step(CFG with the optimised embeddings) - prev_step(CFG with the original ones), and the walk continues
    from the optimised step.  context4 rows = [unc_src, unc_tgt, cond_src, cond_tgt].  Returns the list of [2, 4, h, w] offsets."""
    n = len(timesteps)
    ratio = len(ac) // n
    nrow = context4.shape[0] // 2
    uncond, cond = context4[:nrow].clone(), context4[nrow:]
    latent_cur = torch.cat([ddim_latents[-1]] * nrow)
    out = []
    for i in range(n):
        t = int(timesteps[i])
        a_t, a_p = prev_alphas(ac, final, t, ratio)
        latent_prev = ddim_latents[len(ddim_latents) - i - 2]
        losses = []
        if num_inner_steps != 0:
            uncond = uncond.clone().detach()
            lr = 1e-2 * (1.0 - i / 100.0)
            m, v = torch.zeros_like(uncond), torch.zeros_like(uncond)
            for j in range(num_inner_steps):
                u = uncond.clone().detach().requires_grad_(True)
                with torch.enable_grad():
                    eps = unet_fn(torch.cat([latent_cur] * 2), t, torch.cat([u, cond]), None)
                    eu, ec = eps.chunk(2)
                    rec = ddim_move(latent_cur, eu + guidance_scale * (ec - eu), float(a_t), float(a_p))
                    loss = F.mse_loss(rec[:1], latent_prev)
                g, = torch.autograd.grad(loss, u)
                uncond, m, v = _adam_update(uncond, g, m, v, j + 1, lr)
                losses.append(float(loss.detach()))
                if losses[-1] < epsilon + i * 2e-5:
                    break
        with torch.no_grad():
            eu, ec = unet_fn(torch.cat([latent_cur] * 2), t, context4, None).chunk(2)
            plain = ddim_move(latent_cur, eu + guidance_scale * (ec - eu), float(a_t), float(a_p))
            eu, ec = unet_fn(torch.cat([latent_cur] * 2), t, torch.cat([uncond, cond]), None).chunk(2)
            opt = ddim_move(latent_cur, eu + guidance_scale * (ec - eu), float(a_t), float(a_p))
        out.append((opt - plain).detach())
        latent_cur = plain + out[-1]
        if trace is not None:
            trace.append((i, losses))
    return out


def image2latent(vae_encode_mean_fn, image_u8):
    """utils/utils.py:68-80"""
    x = torch.from_numpy(image_u8).float() / 127.5 - 1
    x = x.permute(2, 0, 1).unsqueeze(0)
    return vae_encode_mean_fn(x) * 0.18215


def latent2image(vae_decode_fn, latents):
    """utils/utils.py:58-66"""
    x = vae_decode_fn(1 / 0.18215 * latents)
    x = (x / 2 + 0.5).clamp(0, 1)
    x = x.permute(0, 2, 3, 1).numpy()
    return (x * 255).astype(np.uint8)

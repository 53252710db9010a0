// Latent-space step kernels: DDIM invert / denoise update, classifier-free guidance, the direct-inversion "3 lines",
// and LocalBlend. fp32 NCHW latents, coalesced, wavefront-uniform scalars; every multiply/add is an explicit
// round-to-nearest op in the reference's order (no FMA contraction), so these kernels are bit-exact against the
// reference formulas given the same eps.
//
// Reference: DirectInversion.next_step models/p2p/inversion.py:262-270, prev_step :247-260,
//   DDIMSchedulerDev.step models/p2p/scheduler_dev.py:38-95 (eta = 0, epsilon prediction, no clipping),
//   CFG + offset lines inversion.py:383-389 and p2p_guidance_forward.py:110-114,
//   LocalBlend.__call__/get_mask models/p2p/attention_control.py:97-121.
#include "ops.h"

// The reference evaluates each multiply / add / divide as a separately rounded fp32 op (eager PyTorch).  hipcc would
// contract a*b+c into one FMA (single rounding) by default, so contraction is switched off for this file.
#pragma clang fp contract(off)

__device__ __forceinline__ float ddim_update(float x, float e, float sqrt_a_from, float sqrt_b_from, float sqrt_a_to,
                                             float sqrt_b_to) {
  // pred_x0 = (x - sqrt(1-a_from) * e) / sqrt(a_from);  dir = sqrt(1-a_to) * e;  out = sqrt(a_to) * pred_x0 + dir
  float x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(sqrt_b_from, e)), sqrt_a_from);
  float dir = __fmul_rn(sqrt_b_to, e);
  return __fadd_rn(__fmul_rn(sqrt_a_to, x0), dir);
}

__global__ void ddim_move_kernel(const float* __restrict__ x, const float* __restrict__ eps, float sa_f, float sb_f, float sa_t,
                                 float sb_t, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ddim_update(x[i], eps[i], sa_f, sb_f, sa_t, sb_t);
}

int launch_ddim_move(const float* x, const float* eps, float a_from, float a_to, size_t n, float* out, hipStream_t st) {
  // torch computes `t ** 0.5` on 0-dim fp32 tensors: correctly rounded fp32 sqrt of fp32 operands
  float sa_f = sqrtf(a_from), sb_f = sqrtf(1.0f - a_from), sa_t = sqrtf(a_to), sb_t = sqrtf(1.0f - a_to);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  ddim_move_kernel<<<blocks, 256, 0, st>>>(x, eps, sa_f, sb_f, sa_t, sb_t, n, out);
  return (int)hipGetLastError();
}

// DDIMSchedulerDev.step with the reconstruction pull (scheduler_dev.py:68-76), level-1 form: ref / mask already expanded to the sample's
// shape.  pred_x0 = (x - sqrt(1-a_t) e) / sqrt(a_t);  pred_x0 -= lr * (pred_x0 - ref) [* mask];  prev = sqrt(a_p) pred_x0 + sqrt(1-a_p) e.
// pred_x0_out (nullable) receives the pulled pred_original_sample (the reference returns it).
__global__ void ddim_prev_recon_kernel(const float* __restrict__ x, const float* __restrict__ eps, float sa_f, float sb_f, float sa_t, float sb_t,
                                       const float* __restrict__ ref, float lr, const float* __restrict__ mask, size_t n,
                                       float* __restrict__ out, float* __restrict__ x0_out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = eps[i];
    float x0 = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(sb_f, e)), sa_f);
    if (ref) {
      float pull = __fmul_rn(lr, __fsub_rn(x0, ref[i]));
      if (mask) pull = __fmul_rn(pull, mask[i]);
      x0 = __fsub_rn(x0, pull);
    }
    if (x0_out) x0_out[i] = x0;
    out[i] = __fadd_rn(__fmul_rn(sa_t, x0), __fmul_rn(sb_t, e));
  }
}
int launch_ddim_prev_recon(const float* x, const float* eps, float a_from, float a_to, const float* ref, float lr, const float* mask, size_t n,
                           float* out, float* x0_out, hipStream_t st) {
  float sa_f = sqrtf(a_from), sb_f = sqrtf(1.0f - a_from), sa_t = sqrtf(a_to), sb_t = sqrtf(1.0f - a_to);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  ddim_prev_recon_kernel<<<blocks, 256, 0, st>>>(x, eps, sa_f, sb_f, sa_t, sb_t, ref, lr, mask, n, out, x0_out);
  return (int)hipGetLastError();
}

// proximal guidance (proximal_guidance_forward.py:39-62): score_delta -= clamp(score_delta, -thr, thr); 'l1' then shrinks the
// survivors by thr once more on each side
__device__ __forceinline__ float prox_shrink(float d, float th, int mode) {
  d = __fsub_rn(d, fminf(fmaxf(d, -th), th));
  if (mode == 2) {
    if (d > 0.f) d = __fsub_rn(d, th);
    if (d < 0.f) d = __fadd_rn(d, th);
  }
  return d;
}

// eps: [nimg][2R][E] (first R rows unconditional, next R conditional); x: [nimg][R][E]
//   e   = eps_u + g * (eps_c - eps_u)
//   prev = ddim(x, e)
//   target != null  (offset_calculate):  loss = (target[img] - prev) * oscale ; offset_out = loss ; x_out = prev + loss
//   else noise_loss != null            :  x_out = prev + noise_loss[img][r]  for r < offset_rows, prev otherwise
// recon_ref != null (reconstruction guidance, proximal_guidance_forward.py:48-51,60-72 + DDIMSchedulerDev.step scheduler_dev.py:68-76):
//   mask_edit = |shrunk delta| > thr, dilated by a (2*dil+1)^2 max-pool over each [h][w] plane (in-bounds neighbours only);
//   pred_x0 -= recon_lr * (pred_x0 - ref[img]) * (1 - mask_edit)   before the step's second half
// inv_ref != null (inversion guidance, proximal_guidance_forward.py:73-75): the step's RESULT is pulled towards the inversion trajectory
//   outside the same mask: x_out -= recon_lr * (x_out - inv_ref[img]) * (1 - mask_edit), inv_ref = x*_{t-1} for both rows of the image
__global__ void cfg_ddim_prev_kernel(const float* __restrict__ eps, const float* __restrict__ x, int nimg, int R, size_t E, float g,
                                     float sa_f, float sb_f, float sa_t, float sb_t, const float* __restrict__ noise_loss,
                                     int offset_rows, const float* __restrict__ target, float oscale,
                                     float* __restrict__ offset_out, float* __restrict__ x_out, const float* __restrict__ prox_thr,
                                     int prox_mode, const float* __restrict__ recon_ref, float recon_lr, int dil, int lat_h, int lat_w,
                                     const float* __restrict__ inv_ref) {
  const size_t total = (size_t)nimg * R * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t e_idx = i % E;
    size_t ir = i / E;
    int r = (int)(ir % R);
    int img = (int)(ir / R);
    const float* eu_row = eps + ((size_t)img * 2 * R + r) * E;
    const float* ec_row = eps + ((size_t)img * 2 * R + R + r) * E;
    float eu = eu_row[e_idx];
    float ec = ec_row[e_idx];
    float d = __fsub_rn(ec, eu);
    const float th = prox_mode ? prox_thr[img] : 0.f;
    if (prox_mode) d = prox_shrink(d, th, prox_mode);
    float e = __fadd_rn(eu, __fmul_rn(g, d));
    float prev;
    if ((recon_ref || inv_ref) && prox_mode) {
      float mask_edit = fabsf(d) > th ? 1.f : 0.f;
      if (dil > 0 && mask_edit == 0.f) {
        const int hw = lat_h * lat_w;
        const int pl = (int)(e_idx / hw), py = (int)(e_idx % hw) / lat_w, px = (int)(e_idx % hw) % lat_w;
        for (int dy = -dil; dy <= dil && mask_edit == 0.f; ++dy)
          for (int dx = -dil; dx <= dil; ++dx) {
            const int yy = py + dy, xx = px + dx;
            if (yy < 0 || yy >= lat_h || xx < 0 || xx >= lat_w) continue;
            const size_t j = (size_t)pl * hw + (size_t)yy * lat_w + xx;
            const float dn = prox_shrink(__fsub_rn(ec_row[j], eu_row[j]), th, prox_mode);
            if (fabsf(dn) > th) { mask_edit = 1.f; break; }
          }
      }
      const float recon_mask = __fsub_rn(1.f, mask_edit);
      if (recon_ref) {
        float x0 = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(sb_f, e)), sa_f);
        x0 = __fsub_rn(x0, __fmul_rn(__fmul_rn(recon_lr, __fsub_rn(x0, recon_ref[(size_t)img * E + e_idx])), recon_mask));
        prev = __fadd_rn(__fmul_rn(sa_t, x0), __fmul_rn(sb_t, e));
      } else {
        prev = ddim_update(x[i], e, sa_f, sb_f, sa_t, sb_t);
      }
      if (inv_ref) prev = __fsub_rn(prev, __fmul_rn(__fmul_rn(recon_lr, __fsub_rn(prev, inv_ref[(size_t)img * E + e_idx])), recon_mask));
    } else {
      prev = ddim_update(x[i], e, sa_f, sb_f, sa_t, sb_t);
    }
    float outv = prev;
    if (target) {
      // loss = (x*_{t-1} - prev) * scale: scale is 1 on the paper's path, `scale` / 0-or-1 in the not_full / skip_step ablations
      // (inversion.py:491-492, 512-515)
      float loss = __fmul_rn(__fsub_rn(target[(size_t)img * E + e_idx], prev), oscale);
      offset_out[i] = loss;
      outv = __fadd_rn(prev, loss);
    } else if (noise_loss && r < offset_rows) {
      outv = __fadd_rn(prev, noise_loss[i]);
    }
    x_out[i] = outv;
  }
}

int launch_cfg_ddim_prev(const float* eps, const float* x, int nimg, int rows_per_img, size_t row_elems, float gscale, float a_t,
                         float a_prev, const float* noise_loss, int offset_rows, const float* target, float offset_scale,
                         float* offset_out, float* x_out, hipStream_t st, const float* prox_thr, int prox_mode, const float* recon_ref,
                         float recon_lr, int dilate, int lat_h, int lat_w, const float* inv_ref) {
  float sa_f = sqrtf(a_t), sb_f = sqrtf(1.0f - a_t), sa_t = sqrtf(a_prev), sb_t = sqrtf(1.0f - a_prev);
  size_t total = (size_t)nimg * rows_per_img * row_elems;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  if ((recon_ref || inv_ref) && (lat_h <= 0 || lat_w <= 0 || row_elems % ((size_t)lat_h * lat_w))) return -3;
  cfg_ddim_prev_kernel<<<blocks, 256, 0, st>>>(eps, x, nimg, rows_per_img, row_elems, gscale, sa_f, sb_f, sa_t, sb_t, noise_loss,
                                               offset_rows, target, offset_scale, offset_out, x_out, prox_thr, prox_mode, recon_ref,
                                               recon_lr, dilate, lat_h, lat_w, inv_ref);
  return (int)hipGetLastError();
}

// torch.quantile(|eps_c - eps_u|, q) over all R rows of one image (proximal_guidance_forward.py:41,53: the threshold of the
// proximal step), default 'linear' interpolation: sort, pos = q * (n - 1), lerp(x[floor], x[floor + 1], frac).  One block per
// image; the n <= 32768 magnitudes are sorted in LDS (bitonic, padded with +inf to a power of two).
__global__ void __launch_bounds__(1024) quantile_abs_diff_kernel(const float* __restrict__ eps, int R, size_t E, float q, int npow2,
                                                                 float* __restrict__ thr_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s = reinterpret_cast<float*>(smem_raw);
  const int img = blockIdx.x, tid = threadIdx.x;
  const int n = (int)(R * E);
  const float* base = eps + (size_t)img * 2 * R * E;
  for (int i = tid; i < npow2; i += blockDim.x)
    s[i] = i < n ? fabsf(__fsub_rn(base[(size_t)R * E + i], base[i])) : INFINITY;
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float a = s[i], b = s[l];
          if ((a > b) == up) { s[i] = b; s[l] = a; }
        }
      }
      __syncthreads();
    }
  if (tid == 0) {
    const float pos = __fmul_rn(q, (float)(n - 1));
    const int lo = (int)floorf(pos);
    const int hi = lo + 1 < n ? lo + 1 : n - 1;
    const float w = __fsub_rn(pos, (float)lo);
    const float a = s[lo], b = s[hi];
    // torch.lerp: a + w * (b - a) for w < 0.5, b - (b - a) * (1 - w) otherwise
    thr_out[img] = w < 0.5f ? __fadd_rn(a, __fmul_rn(w, __fsub_rn(b, a))) : __fsub_rn(b, __fmul_rn(__fsub_rn(b, a), __fsub_rn(1.f, w)));
  }
}

int launch_quantile_abs_diff(const float* eps, int nimg, int rows_per_img, size_t row_elems, float q, float* thr_out, hipStream_t st) {
  const size_t n = (size_t)rows_per_img * row_elems;
  int npow2 = 1;
  while ((size_t)npow2 < n) npow2 <<= 1;
  if (npow2 > 32768) return -6;
  static DeviceOnce attr_once;
  if (int r = once_per_device(attr_once, [&]() { return (int)hipFuncSetAttribute((const void*)quantile_abs_diff_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4); })) return r;
  quantile_abs_diff_kernel<<<nimg, 1024, (size_t)npow2 * sizeof(float), st>>>(eps, rows_per_img, row_elems, q, npow2, thr_out);
  return (int)hipGetLastError();
}

__global__ void fill_f32_kernel(float* p, int n, float v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int launch_fill_f32(float* p, int n, float v, hipStream_t st) {
  fill_f32_kernel<<<(n + 255) / 256, 256, 0, st>>>(p, n, v);
  return (int)hipGetLastError();
}

// One block per image. lb_acc: [nimg][nslots][planes][map_hw*map_hw] accumulated (summed over steps) selector-weighted maps; planes 0, 1 =
// the blend words of the source / target prompt, planes 2, 3 (planes == 4) = the substruct words.
// blend maps   -> mean over slots -> 3x3 max-pool (stride 1, pad 1) -> nearest resize to lat_hw -> / max -> > th      (get_mask, use_pool)
// substruct    -> mean over slots ->              (no pooling)      -> nearest resize          -> / max -> > th_sub  (attention_control.py:97-118)
// mask_tgt = (mask_src | mask_tgt) & ~(sub_src | sub_tgt);  x_tgt = x_src + mask_tgt * (x_tgt - x_src)   (latents [nimg][2][C][lat_hw^2])
__global__ void __launch_bounds__(256) local_blend_kernel(const float* __restrict__ lb_acc, int nslots, int mhw, int lhw, int C,
                                                          float th, float* __restrict__ latents, int planes, float th_sub) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int MP = mhw * mhw;
  float* s_map = reinterpret_cast<float*>(smem_raw);  // [4][MP]   mean maps
  float* s_pool = s_map + 4 * MP;                     // [4][MP]   pooled (planes 0, 1) / copied (planes 2, 3)
  float* s_red = s_pool + 4 * MP;                     // [4][256]
  const int img = blockIdx.x, tid = threadIdx.x;
  for (int idx = tid; idx < planes * MP; idx += blockDim.x) {
    int which = idx / MP, pix = idx - which * MP;
    float s = 0.f;
    for (int sl = 0; sl < nslots; ++sl) s += lb_acc[(((size_t)img * nslots + sl) * planes + which) * MP + pix];
    s_map[idx] = s / (float)nslots;
  }
  __syncthreads();
  float lmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int idx = tid; idx < planes * MP; idx += blockDim.x) {
    int which = idx / MP, pix = idx - which * MP;
    int y = pix / mhw, x = pix - y * mhw;
    float m = -INFINITY;
    if (which < 2) {
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int yy = y + dy, xx = x + dx;
          if (yy >= 0 && yy < mhw && xx >= 0 && xx < mhw) m = fmaxf(m, s_map[which * MP + yy * mhw + xx]);
        }
    } else {
      m = s_map[idx];
    }
    s_pool[idx] = m;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w == which) lmax[w] = fmaxf(lmax[w], m);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) s_red[w * 256 + tid] = lmax[w];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int w = 0; w < 4; ++w) s_red[w * 256 + tid] = fmaxf(s_red[w * 256 + tid], s_red[w * 256 + tid + off]);
    }
    __syncthreads();
  }
  const float mx0 = s_red[0], mx1 = s_red[256], mx2 = s_red[512], mx3 = s_red[768];
  const int LP = lhw * lhw;
  float* xs = latents + (size_t)img * 2 * C * LP;
  float* xt = xs + (size_t)C * LP;
  for (int pix = tid; pix < LP; pix += blockDim.x) {
    int y = pix / lhw, x = pix - y * lhw;
    // torch 'nearest': src = floor(dst * in / out)
    int sy = (int)floorf((float)y * ((float)mhw / (float)lhw)), sx = (int)floorf((float)x * ((float)mhw / (float)lhw));
    if (sy > mhw - 1) sy = mhw - 1;
    if (sx > mhw - 1) sx = mhw - 1;
    float v0 = __fdiv_rn(s_pool[sy * mhw + sx], mx0);
    float v1 = __fdiv_rn(s_pool[MP + sy * mhw + sx], mx1);
    bool m = (v0 > th) || (v1 > th);
    if (planes == 4) {
      float u0 = __fdiv_rn(s_pool[2 * MP + sy * mhw + sx], mx2);
      float u1 = __fdiv_rn(s_pool[3 * MP + sy * mhw + sx], mx3);
      m = m && !((u0 > th_sub) || (u1 > th_sub));
    }
    float mf = m ? 1.f : 0.f;
    for (int c = 0; c < C; ++c) {
      float a = xs[(size_t)c * LP + pix], b = xt[(size_t)c * LP + pix];
      xt[(size_t)c * LP + pix] = __fadd_rn(a, __fmul_rn(mf, __fsub_rn(b, a)));
    }
  }
}

int launch_local_blend(const float* lb_acc, int nslots, int map_hw, int lat_hw, int C, float th, float* latents, int nimg,
                       hipStream_t st, int planes, float th_sub) {
  if (planes != 2 && planes != 4) return -3;
  size_t lds = (size_t)(8 * map_hw * map_hw + 1024) * sizeof(float);
  local_blend_kernel<<<nimg, 256, lds, st>>>(lb_acc, nslots, map_hw, lat_hw, C, th, latents, planes, th_sub);
  return (int)hipGetLastError();
}

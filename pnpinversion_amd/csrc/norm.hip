// Bandwidth-bound kernels: GroupNorm(+SiLU), LayerNorm, GEGLU, row softmax, GEMV, layout/dtype conversions.
// Reference ops restated: torch.nn.GroupNorm (my_diffusers/models/resnet.py:287,296; attention.py:123; unet_2d_condition.py:163),
// torch.nn.LayerNorm (attention.py:186-188), GEGLU (attention.py:323-333), softmax (attention.py:77),
// TimestepEmbedding / time_emb_proj linears (embeddings.py:63-80, resnet.py:292,349).
// All statistics are fp32; 16-byte (8 x fp16) vector accesses, NHWC so that a wavefront reads contiguous channels.
#include <algorithm>
#include "ops.h"

// ------------------------------------------------------------------------------------------------ GroupNorm
// Three launches, all latency-lean:
//   gn_stats    (B x nchunk blocks)  per-chunk partial (sum, sumsq) per group; each thread owns a fixed 8-channel vector and walks
//                                    pixels four at a time (independent loads in flight); fixed-order reduction, no atomics
//   gn_finalize (B blocks)           partials -> per-channel scale = rstd*gamma, shift = beta - mean*scale   ([B][C][2] fp32)
//   gn_apply    (B x napply blocks)  y = x*scale + shift (+SiLU), 16-byte accesses
__global__ void __launch_bounds__(256) gn_stats_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int C1,
                                                       int C2, int HW, int G, int nchunk, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_part = reinterpret_cast<float*>(smem_raw);  // [TP][C][2]
  const int C = C1 + C2, C8 = C >> 3;
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int TC = C8 < 256 ? C8 : 256;
  const int TP = 256 / TC;
  const int tc = threadIdx.x % TC, tp = threadIdx.x / TC;
  const int ppc = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  if (tp < TP) {
    for (int cv = tc; cv < C8; cv += TC) {
      float s[8], q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
      const int c = cv * 8;
      const half_t* src; int ld, cc;
      if (c < C1) { src = x1; ld = C1; cc = c; } else { src = x2; ld = C2; cc = c - C1; }
      const half_t* base = src + (size_t)b * HW * ld + cc;
      int pix = p0 + tp;
      for (; pix + 3 * TP < p1; pix += 4 * TP) {
        half8 v0 = ldg_half8(base + (size_t)pix * ld);
        half8 v1 = ldg_half8(base + (size_t)(pix + TP) * ld);
        half8 v2 = ldg_half8(base + (size_t)(pix + 2 * TP) * ld);
        half8 v3 = ldg_half8(base + (size_t)(pix + 3 * TP) * ld);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f0 = (float)v0[j], f1 = (float)v1[j], f2 = (float)v2[j], f3 = (float)v3[j];
          s[j] += (f0 + f1) + (f2 + f3);
          q[j] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
        }
      }
      for (; pix < p1; pix += TP) {
        half8 v = ldg_half8(base + (size_t)pix * ld);
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = (float)v[j]; s[j] += f; q[j] += f * f; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s_part[((size_t)tp * C + c + j) * 2 + 0] = s[j];
        s_part[((size_t)tp * C + c + j) * 2 + 1] = q[j];
      }
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int t = 0; t < TP; ++t)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        s += s_part[((size_t)t * C + c) * 2 + 0];
        q += s_part[((size_t)t * C + c) * 2 + 1];
      }
    float* dst = partial + (((size_t)b * nchunk + chunk) * G + g) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

// One block per batch row: 8 lanes per group sum the chunk partials (fixed order), then scale/shift for every channel.
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partial, int C, int HW, int G, int nchunk,
                                                          float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ ss) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int g = g0 + (tid >> 3), sub = tid & 7;
    float s = 0.f, q = 0.f;
    if (g < G)
      for (int ch = sub; ch < nchunk; ch += 8) {
        const float* src = partial + (((size_t)b * nchunk + ch) * G + g) * 2;
        s += src[0];
        q += src[1];
      }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
    if (g < G && sub == 0) {
      const float n = (float)HW * (float)cpg;
      float mean = s / n;
      float var = q / n - mean * mean;
      var = var > 0.f ? var : 0.f;
      s_mean[g] = mean;
      s_rstd[g] = rsqrtf(var + eps);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    int g = c / cpg;
    float sc = s_rstd[g] * gamma[c];
    ss[((size_t)b * C + c) * 2 + 0] = sc;
    ss[((size_t)b * C + c) * 2 + 1] = beta[c] - s_mean[g] * sc;
  }
}

// y = x * scale + shift (+ SiLU).  A thread owns one 8-channel vector position (its 16 scale / shift values stay in registers)
// and walks pixels, four independent 16-byte loads in flight; a block covers `ppb` consecutive pixels of one batch item, so a
// wavefront reads and writes whole contiguous rows.  No LDS, no barrier.
__global__ void __launch_bounds__(256) gn_apply_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int C1,
                                                       int C2, int HW, int ppb, const float* __restrict__ ss, int silu,
                                                       half_t* __restrict__ out) {
  const int C = C1 + C2, C8 = C >> 3;
  const int CL = C8 < 256 ? C8 : 256, TP = 256 / CL;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int tc = tid % CL, tp = tid / CL;
  if (tp >= TP) return;
  const int p0 = blockIdx.y * ppb, p1 = min(HW, p0 + ppb);
  for (int cv = tc; cv < C8; cv += CL) {
    const int c = cv * 8;
    float sc[8], sh[8];
    {
      const floatx4* sp = reinterpret_cast<const floatx4*>(ss + ((size_t)b * C + c) * 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) { floatx4 v = sp[j]; sc[2 * j] = v[0]; sh[2 * j] = v[1]; sc[2 * j + 1] = v[2]; sh[2 * j + 1] = v[3]; }
    }
    const half_t* src; int ld, cc;
    if (c < C1) { src = x1; ld = C1; cc = c; } else { src = x2; ld = C2; cc = c - C1; }
    const half_t* base = src + (size_t)b * HW * ld + cc;
    half_t* obase = out + (size_t)b * HW * C + c;
    auto xf = [&](half8 v) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)v[j] * sc[j] + sh[j];
        if (silu) f = silu_f(f);
        o[j] = (half_t)f;
      }
      return o;
    };
    int pix = p0 + tp;
    for (; pix + 3 * TP < p1; pix += 4 * TP) {
      half8 v0 = ldg_half8(base + (size_t)pix * ld);
      half8 v1 = ldg_half8(base + (size_t)(pix + TP) * ld);
      half8 v2 = ldg_half8(base + (size_t)(pix + 2 * TP) * ld);
      half8 v3 = ldg_half8(base + (size_t)(pix + 3 * TP) * ld);
      *reinterpret_cast<half8*>(obase + (size_t)pix * C) = xf(v0);
      *reinterpret_cast<half8*>(obase + (size_t)(pix + TP) * C) = xf(v1);
      *reinterpret_cast<half8*>(obase + (size_t)(pix + 2 * TP) * C) = xf(v2);
      *reinterpret_cast<half8*>(obase + (size_t)(pix + 3 * TP) * C) = xf(v3);
    }
    for (; pix < p1; pix += TP) *reinterpret_cast<half8*>(obase + (size_t)pix * C) = xf(ldg_half8(base + (size_t)pix * ld));
  }
}

// Small feature maps (16x16 / 8x8 levels): one block per (batch item, group) does everything in ONE launch -- the group's
// HW x cpg slice is parked in LDS between the statistics pass and the apply pass.  These tensors are < 1 MB; three dependent
// launches were pure latency.
// largest (sample, group) slice of gn_small_kernel: its registers hold the slice.  (Round 5: raising it to the 64 x 64 level's 40 960
// elements for launches of fewer than 256 blocks -- one launch instead of two or three on the one-row forwards -- measured neutral,
// 4.556 vs 4.548 ms per one-row forward: not kept.)
static constexpr int GN_SMALL_ELEMS = 24576;
// SLAB: the first source is not a tensor yet but the split-K slabs of the GEMM that produces it (GnSlab, ops.h): the block sums the
// slabs of its slice in slab order, adds bias and residual and rounds to fp16 -- operation for operation what splitk_reduce_vec_kernel
// does, so the statistics and the output are bit-identical to the two-launch path -- and stores that tensor only if somebody else reads it.
template <int NT, bool SLAB = false>
__global__ void __launch_bounds__(NT) gn_small_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int C1, int C2,
                                                       int HW, int G, float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int silu, half_t* __restrict__ out, GnSlab sl = GnSlab()) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half4* s_x = reinterpret_cast<half4*>(smem_raw);     // [HW * cpg / 4]
  __shared__ float s_s[NT / 64], s_q[NT / 64];
  const int C = C1 + C2, cpg = C / G, v4 = cpg >> 2;
  const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int nvec = HW * v4;
  floatx4* s_ga = reinterpret_cast<floatx4*>(smem_raw + (((size_t)nvec * sizeof(half4) + 15) & ~(size_t)15));   // the group's gamma, then beta
  floatx4* s_be = s_ga + v4;
  constexpr int GPT = 1024 / NT;               // cpg <= 1024: affine parameters per thread, loaded (clamped) ahead of the data loads
  float ga_r[GPT], be_r[GPT];
#pragma unroll
  for (int u = 0; u < GPT; ++u) {
    const int cl = min(tid + u * NT, cpg - 1);
    ga_r[u] = gamma[g * cpg + cl];
    be_r[u] = beta[g * cpg + cl];
  }
  float s = 0.f, q = 0.f;
  // Every slice of the thread in flight at once: MAXV clamped (unpredicated) 8-byte loads issued back to back in straight-line code,
  // then consumed in index order -- the one-load-one-wait loop this replaces was a chain of up to six HBM / L2 round trips per
  // thread in a kernel that is pure latency (a loop around batches makes the compiler drain the counter at the loop header).
  constexpr int MAXV = GN_SMALL_ELEMS / 4 / NT;
  half4 val[MAXV];
  if constexpr (!SLAB) {
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int idx = min(tid + u * NT, nvec - 1);
      const int pix = idx / v4, v = idx - pix * v4;
      const int c = g * cpg + 4 * v;
      const bool first = c < C1;            // selects on the operands, one address computation: no divergent control flow between the loads
      const half_t* sb = first ? x1 : x2;
      const int ld = first ? C1 : C2, cc = first ? c : c - C1;
      val[u] = *reinterpret_cast<const half4*>(sb + ((size_t)b * HW + pix) * ld + cc);
    }
  } else {
    floatx4 acc[MAXV];
    size_t eoff[MAXV];                      // element offset of the vector in [M][C1] (slab, bias-free sum_out; ldo == C1)
    bool isl[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int idx = min(tid + u * NT, nvec - 1);
      const int pix = idx / v4, v = idx - pix * v4;
      const int c = g * cpg + 4 * v;
      isl[u] = c < C1;
      eoff[u] = ((size_t)b * HW + pix) * C1 + (isl[u] ? c : 0);
      acc[u] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (!isl[u]) val[u] = *reinterpret_cast<const half4*>(x2 + ((size_t)b * HW + pix) * C2 + (c - C1));
    }
#pragma unroll 2
    for (int z = 0; z < sl.splitk; ++z) {
      const float* sz = sl.slab + (size_t)z * sl.stride;
#pragma unroll
      for (int u = 0; u < MAXV; ++u)
        if (isl[u]) acc[u] += *reinterpret_cast<const floatx4*>(sz + eoff[u]);
    }
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      if (!isl[u]) continue;
      const int idx = min(tid + u * NT, nvec - 1);
      const int pix = idx / v4, v = idx - pix * v4;
      const int c = g * cpg + 4 * v;
      floatx4 w = acc[u];
      if (sl.bias) w += *reinterpret_cast<const floatx4*>(sl.bias + c);
      if (sl.res) {
        const half4 r4 = *reinterpret_cast<const half4*>(sl.res + ((size_t)b * HW + pix) * sl.ldres + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] += (float)r4[j];
      }
      const half4 h = {(half_t)w[0], (half_t)w[1], (half_t)w[2], (half_t)w[3]};
      val[u] = h;
      if (sl.sum_out && tid + u * NT < nvec) *reinterpret_cast<half4*>(sl.sum_out + eoff[u]) = h;
    }
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int idx = tid + u * NT;
    const bool ok = idx < nvec;
    if (ok) s_x[idx] = val[u];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // unconditional use (a clamped duplicate adds exact zeros): keeps every load ahead of the first wait
      const float f = ok ? (float)val[u][j] : 0.f;
      s += f; q += f * f;
    }
  }
#pragma unroll
  for (int u = 0; u < GPT; ++u) {              // read back as one float4 per 4-channel vector
    const int cl = tid + u * NT;
    if (cl < cpg) { reinterpret_cast<float*>(s_ga)[cl] = ga_r[u]; reinterpret_cast<float*>(s_be)[cl] = be_r[u]; }
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((tid & 63) == 0) { s_s[tid >> 6] = s; s_q[tid >> 6] = q; }
  __syncthreads();
  s = 0.f; q = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { s += s_s[w]; q += s_q[w]; }
  const float n = (float)HW * (float)cpg;
  const float mean = s / n;
  float var = q / n - mean * mean;
  var = var > 0.f ? var : 0.f;
  const float rstd = rsqrtf(var + eps);
  for (int idx = tid; idx < nvec; idx += NT) {
    const int pix = idx / v4, v = idx - pix * v4;
    const int c = g * cpg + 4 * v;
    const half4 val = s_x[idx];
    const floatx4 ga = s_ga[v], be = s_be[v];
    half4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sc = rstd * ga[j];
      float f = ((float)val[j] - mean) * sc + be[j];
      if (silu) f = silu_f(f);
      o[j] = (half_t)f;
    }
    *reinterpret_cast<half4*>(out + ((size_t)b * HW + pix) * C + c) = o;
  }
}

static int gn_small_max() {     // elements per (sample, group) slice handled by the one-launch kernel (PNPI_GN_SMALL_MAX lowers it)
  static const int v = getenv("PNPI_GN_SMALL_MAX") ? std::min(atoi(getenv("PNPI_GN_SMALL_MAX")), GN_SMALL_ELEMS) : GN_SMALL_ELEMS;
  return v;
}
// B (optional): at 48 rows and more (1536 blocks) only slices up to 8 192 elements take the one-launch kernel -- with six blocks per CU
// the streaming statistics / finalize / apply passes are faster on the larger slices (96-row forward 94.2 -> 93.4 ms, no difference at 12 rows;
// round 5, PNPI_GN_SMALL_MAX sweep)
static bool gn_small_ok(int C1, int C2, int HW, int G, int B = 0) {
  const int C = C1 + C2, cpg = C / G;
  const size_t lim = (long)B * G >= 1536 ? (size_t)std::min(gn_small_max(), 8192) : (size_t)gn_small_max();
  return (cpg % 4 == 0) && (C1 % 4 == 0) && cpg <= 1024 && ((size_t)HW * cpg <= lim);   // gamma / beta: 8 KB of LDS at most
}
static int launch_gn_small(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                           const float* beta, int silu, half_t* out, hipStream_t st) {
  const int cpg = (C1 + C2) / G;
  static DeviceOnce attr_once;
  // the launch asks for HW * cpg * 2 + cpg * 8 + 16 bytes (slice + gamma / beta + alignment slack): <= gn_small_max() * 2 + 8192 + 16
  if (int r = once_per_device(attr_once, [&]() {
        const int lim = gn_small_max() * 2 + 8192 + 16;
        int e = (int)hipFuncSetAttribute((const void*)gn_small_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        return e ? e : (int)hipFuncSetAttribute((const void*)gn_small_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
      })) return r;
  // 1024-thread blocks: each (sample, group) slice is a latency chain load -> reduce -> normalise -> store, and four times
  // the lanes per slice shorten it at every row count measured (1 row: 16.2 -> 12.7 us per launch; 12 rows: 23.0 -> 20.7)
  static const int wide_below = getenv("PNPI_GN_WIDE_BELOW") ? atoi(getenv("PNPI_GN_WIDE_BELOW")) : (1 << 30);
  if (B * G < wide_below)
    gn_small_kernel<1024><<<dim3(B, G), 1024, (size_t)HW * cpg * sizeof(half_t) + (size_t)cpg * 8 + 16, st>>>(x1, x2, C1, C2, HW, G, eps, gamma, beta, silu, out);
  else
    gn_small_kernel<256><<<dim3(B, G), 256, (size_t)HW * cpg * sizeof(half_t) + (size_t)cpg * 8 + 16, st>>>(x1, x2, C1, C2, HW, G, eps, gamma, beta, silu, out);
  return (int)hipGetLastError();
}

bool groupnorm_slab_ok(int C1, int C2, int HW, int G) { return C1 > 0 && (C2 == 0 || C2 % 4 == 0) && gn_small_ok(C1, C2, HW, G); }
int launch_groupnorm_slab(const GnSlab& sl, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                          const float* beta, int silu, half_t* out, hipStream_t st) {
  const int C = C1 + C2;
  if ((C & 7) || (C1 & 7) || C % G || G > 64 || !groupnorm_slab_ok(C1, C2, HW, G) || sl.splitk < 2 || !sl.slab) return -3;
  if (((uintptr_t)sl.slab & 15) || (sl.bias && ((uintptr_t)sl.bias & 15)) || (sl.res && (((uintptr_t)sl.res & 7) || (sl.ldres & 3))) ||
      (sl.sum_out && ((uintptr_t)sl.sum_out & 7)) || (sl.stride & 3))
    return -3;
  const int cpg = C / G;
  static DeviceOnce attr_once;
  if (int r = once_per_device(attr_once, [&]() {
        return (int)hipFuncSetAttribute((const void*)gn_small_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gn_small_max() * 2 + 8192 + 16);
      })) return r;
  gn_small_kernel<1024, true><<<dim3(B, G), 1024, (size_t)HW * cpg * sizeof(half_t) + (size_t)cpg * 8 + 16, st>>>(nullptr, x2, C1, C2, HW, G, eps, gamma, beta,
                                                                                                          silu, out, sl);
  return (int)hipGetLastError();
}

static int gn_nchunk(int HW) { int n = HW / 64; if (n < 1) n = 1; if (n > 128) n = 128; return n; }
// pixels per apply block: every thread gets >= 8 pixels of its vector position where the map allows, >= ~8 blocks per CU overall
static int gn_ppb(int B, int HW, int C) {
  const int C8 = C >> 3, CL = C8 < 256 ? C8 : 256, TP = 256 / CL;
  int ppb = 8 * TP;
  while (ppb > TP && (long)B * ((HW + ppb - 1) / ppb) < 2048) ppb >>= 1;
  if (ppb < TP) ppb = TP;
  return ppb;
}

// partial: fp32 scratch of at least B * (nchunk * G * 2 + C * 2) floats
int launch_groupnorm(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                     const float* beta, int silu, half_t* out, float* partial, hipStream_t st) {
  const int C = C1 + C2;
  if ((C & 7) || (C1 & 7) || C % G || G > 64) return -3;
  if (gn_small_ok(C1, C2, HW, G, B)) return launch_gn_small(x1, x2, C1, C2, B, HW, G, eps, gamma, beta, silu, out, st);
  const int C8 = C >> 3;
  const int TC = C8 < 256 ? C8 : 256, TP = 256 / TC;
  const int nchunk = gn_nchunk(HW);
  float* ss = partial + (size_t)B * nchunk * G * 2;
  size_t lds1 = (size_t)TP * C * 2 * sizeof(float);
  gn_stats_kernel<<<dim3(B, nchunk), 256, lds1, st>>>(x1, x2, C1, C2, HW, G, nchunk, partial);
  gn_finalize_kernel<<<B, 256, 0, st>>>(partial, C, HW, G, nchunk, eps, gamma, beta, ss);
  const int ppb = gn_ppb(B, HW, C);
  gn_apply_kernel<<<dim3(B, (HW + ppb - 1) / ppb), 256, 0, st>>>(x1, x2, C1, C2, HW, ppb, ss, silu, out);
  return (int)hipGetLastError();
}

// Finalize from per-(m-tile, channel) partial sums written by the producer GEMM epilogues (two-source concat allowed).
// One block per (batch item, group): cpg * tiles partial pairs are summed by 256 threads (fixed order: strided + tree).
__global__ void __launch_bounds__(256) gn_finalize_tiles_kernel(const float* __restrict__ st1, int C1, int tpb1,
                                                                const float* __restrict__ st2, int C2, int tpb2, int HW, int G,
                                                                float eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ ss) {
  __shared__ float s_s[4], s_q[4];
  const int C = C1 + C2, cpg = C / G;
  const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  float s = 0.f, q = 0.f;
  // the group's (channel, m-tile) partial pairs of each source spread over all 256 threads (8-byte loads, channels fastest: coalesced),
  // fixed thread -> pair assignment and a fixed combine tree: deterministic
  const int cg0 = g * cpg, cg1 = cg0 + cpg;
  {
    const int a0 = min(cg0, C1), n1 = min(cg1, C1) - a0;
    for (int i = tid; i < n1 * tpb1; i += 256) {
      const int t = i / n1, cl = i - t * n1;
      const float2 v = *reinterpret_cast<const float2*>(st1 + (((size_t)b * tpb1 + t) * C1 + a0 + cl) * 2);
      s += v.x; q += v.y;
    }
    const int b0 = max(cg0, C1) - C1, n2 = max(cg1, C1) - C1 - b0;
    for (int i = tid; i < n2 * tpb2; i += 256) {
      const int t = i / n2, cl = i - t * n2;
      const float2 v = *reinterpret_cast<const float2*>(st2 + (((size_t)b * tpb2 + t) * C2 + b0 + cl) * 2);
      s += v.x; q += v.y;
    }
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((tid & 63) == 0) { s_s[tid >> 6] = s; s_q[tid >> 6] = q; }
  __syncthreads();
  s = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  q = (s_q[0] + s_q[1]) + (s_q[2] + s_q[3]);
  const float n = (float)HW * (float)cpg;
  const float mean = s / n;
  float var = q / n - mean * mean;
  var = var > 0.f ? var : 0.f;
  const float rstd = rsqrtf(var + eps);
  for (int cl = tid; cl < cpg; cl += 256) {
    const int c = g * cpg + cl;
    const float sc = rstd * gamma[c];
    ss[((size_t)b * C + c) * 2 + 0] = sc;
    ss[((size_t)b * C + c) * 2 + 1] = beta[c] - mean * sc;
  }
}

// One-launch form of finalize + apply for the statistics-fused path: every block first reduces the per-(m-tile, channel) partial sums
// of its batch item to the G group statistics (8 lanes per group, fixed order: strided, then a 3-step butterfly), then applies
// y = (x - mean) * rstd * gamma + beta (+ SiLU) to its pixel range.  The partials are C * tiles * 8 bytes per batch item (82 KB at
// 64 x 64 x 320) and L2-resident, so a few hundred blocks re-reading them cost less than a second dependent launch.
__global__ void __launch_bounds__(256) gn_fin_apply_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2, int C1, int C2,
                                                           int HW, int ppb, const float* __restrict__ st1, int tpb1,
                                                           const float* __restrict__ st2, int tpb2, int G, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                           half_t* __restrict__ out) {
  __shared__ float s_mean[64], s_rstd[64];
  const int C = C1 + C2, C8 = C >> 3, cpg = C / G;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int g = g0 + (tid >> 3), sub = tid & 7;
    float s = 0.f, q = 0.f;
    if (g < G) {
      for (int cl = 0; cl < cpg; ++cl) {
        const int c = g * cpg + cl;
        const float* st; int cs, cc, tpb;
        if (c < C1) { st = st1; cs = C1; cc = c; tpb = tpb1; } else { st = st2; cs = C2; cc = c - C1; tpb = tpb2; }
        const float* base = st + ((size_t)b * tpb * cs + cc) * 2;
        for (int t = sub; t < tpb; t += 8) { s += base[(size_t)t * cs * 2]; q += base[(size_t)t * cs * 2 + 1]; }
      }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
    if (g < G && sub == 0) {
      const float n = (float)HW * (float)cpg;
      const float mean = s / n;
      float var = q / n - mean * mean;
      var = var > 0.f ? var : 0.f;
      s_mean[g] = mean;
      s_rstd[g] = rsqrtf(var + eps);
    }
  }
  __syncthreads();
  const int CL = C8 < 256 ? C8 : 256, TP = 256 / CL;
  const int tc = tid % CL, tp = tid / CL;
  if (tp >= TP) return;
  const int p0 = blockIdx.y * ppb, p1 = min(HW, p0 + ppb);
  for (int cv = tc; cv < C8; cv += CL) {
    const int c = cv * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (c + j) / cpg;
      sc[j] = s_rstd[g] * gamma[c + j];
      sh[j] = beta[c + j] - s_mean[g] * sc[j];
    }
    const half_t* src; int ld, cc;
    if (c < C1) { src = x1; ld = C1; cc = c; } else { src = x2; ld = C2; cc = c - C1; }
    const half_t* base = src + (size_t)b * HW * ld + cc;
    half_t* obase = out + (size_t)b * HW * C + c;
    auto xf = [&](half8 v) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)v[j] * sc[j] + sh[j];
        if (silu) f = silu_f(f);
        o[j] = (half_t)f;
      }
      return o;
    };
    int pix = p0 + tp;
    for (; pix + 3 * TP < p1; pix += 4 * TP) {
      half8 v0 = ldg_half8(base + (size_t)pix * ld);
      half8 v1 = ldg_half8(base + (size_t)(pix + TP) * ld);
      half8 v2 = ldg_half8(base + (size_t)(pix + 2 * TP) * ld);
      half8 v3 = ldg_half8(base + (size_t)(pix + 3 * TP) * ld);
      *reinterpret_cast<half8*>(obase + (size_t)pix * C) = xf(v0);
      *reinterpret_cast<half8*>(obase + (size_t)(pix + TP) * C) = xf(v1);
      *reinterpret_cast<half8*>(obase + (size_t)(pix + 2 * TP) * C) = xf(v2);
      *reinterpret_cast<half8*>(obase + (size_t)(pix + 3 * TP) * C) = xf(v3);
    }
    for (; pix < p1; pix += TP) *reinterpret_cast<half8*>(obase + (size_t)pix * C) = xf(ldg_half8(base + (size_t)pix * ld));
  }
}

static int g_gn_inline_rows = 0;   // statistics-fused GroupNorm: one launch (finalize inside apply) up to this many B*HW rows
void norm_set_tuning_gn_inline_rows(int v) { g_gn_inline_rows = v; }

int launch_groupnorm_fused(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                           const float* beta, int silu, half_t* out, const float* st1, int tpb1, const float* st2, int tpb2,
                           float* scratch, hipStream_t st) {
  const int C = C1 + C2;
  if ((C & 7) || (C1 & 7) || C % G || G > 64) return -3;
  // the one-launch kernel runs one block per (sample, group) -- a few dozen blocks at one row -- yet measured equal to the
  // two-launch path there (5.72 vs 5.75 ms per forward): no threshold by default
  static const int small_min_blocks = getenv("PNPI_GN_SMALL_MIN") ? atoi(getenv("PNPI_GN_SMALL_MIN")) : 0;
  if (gn_small_ok(C1, C2, HW, G, B) && B * G >= small_min_blocks)
    return launch_gn_small(x1, x2, C1, C2, B, HW, G, eps, gamma, beta, silu, out, st);
  if ((long)B * HW <= g_gn_inline_rows) {
    // ~2 blocks per CU, each with enough pixels to amortise its own reduction of the partials
    const int C8 = C >> 3, CL = C8 < 256 ? C8 : 256, TP = 256 / CL;
    int ppb = (int)(((long)B * HW + 511) / 512);
    ppb = ((ppb + TP - 1) / TP) * TP;
    if (ppb < 4 * TP) ppb = 4 * TP;
    gn_fin_apply_kernel<<<dim3(B, (HW + ppb - 1) / ppb), 256, 0, st>>>(x1, x2, C1, C2, HW, ppb, st1, tpb1, st2, x2 ? tpb2 : 1, G, eps, gamma,
                                                                     beta, silu, out);
    return (int)hipGetLastError();
  }
  float* ss = scratch;   // [B][C][2]
  gn_finalize_tiles_kernel<<<dim3(B, G), 256, 0, st>>>(st1, C1, tpb1, st2, C2, tpb2, HW, G, eps, gamma, beta, ss);
  const int ppb = gn_ppb(B, HW, C);
  gn_apply_kernel<<<dim3(B, (HW + ppb - 1) / ppb), 256, 0, st>>>(x1, x2, C1, C2, HW, ppb, ss, silu, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// A wavefront owns RPW token rows at once; the rows stay in registers (C <= 2048) so mean and variance are two exact passes.
// VPL = 16-byte vectors per lane and row.  One row per wave (640 bytes at C = 320) left a single load in flight per lane and the
// kernel latency-bound at a quarter of the HBM rate; with RPW * VPL loads issued back to back (clamped addresses instead of
// predication, masked lanes zeroed afterwards) and RPW independent reduction chains the same arithmetic runs per row -- the
// per-lane and cross-lane summation order is unchanged, so the output bits are too.
template <int VPL, int RPW>
__global__ void __launch_bounds__(256) layernorm_kernel(const half_t* __restrict__ x, int M, int C, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        half_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= M) return;
  const int C8 = C >> 3;
  half8 v[RPW][VPL];
  bool ok[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) ok[i] = lane + 64 * i < C8;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const half_t* src = x + (size_t)min(row0 + r, M - 1) * C;
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[r][i] = ldg_half8(src + min(lane + 64 * i, C8 - 1) * 8);
  }
  float s[RPW], q[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s[r] += (float)v[r][i][j];
      }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) s[r] = wave_sum(s[r]);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float mean = s[r] / (float)C;
    s[r] = mean;
    q[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = (float)v[r][i][j] - mean; q[r] += d * d; }
      }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) q[r] = wave_sum(q[r]);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (!ok[i]) continue;
    const int c0 = (lane + 64 * i) * 8;
    float ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j]; }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (row0 + r >= M) continue;
      const float mean = s[r], rstd = rsqrtf(q[r] / (float)C + eps);
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)(((float)v[r][i][j] - mean) * rstd * ga[j] + be[j]);
      *reinterpret_cast<half8*>(out + (size_t)(row0 + r) * C + c0) = o;
    }
  }
}

int launch_layernorm(const half_t* x, int M, int C, float eps, const float* gamma, const float* beta, half_t* out,
                     hipStream_t st) {
  if ((C & 7) || C > 2048 || M <= 0) return -3;
  const int C8 = C >> 3;
  if (C8 <= 64) layernorm_kernel<1, 4><<<(M + 15) / 16, 256, 0, st>>>(x, M, C, eps, gamma, beta, out);
  else if (C8 <= 128) layernorm_kernel<2, 2><<<(M + 7) / 8, 256, 0, st>>>(x, M, C, eps, gamma, beta, out);
  else layernorm_kernel<4, 1><<<(M + 3) / 4, 256, 0, st>>>(x, M, C, eps, gamma, beta, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ GEGLU
__global__ void __launch_bounds__(256) geglu_kernel(const half_t* __restrict__ x, int M, int I, half_t* __restrict__ out) {
  const int I8 = I >> 3;
  const size_t total = (size_t)M * I8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t m = idx / I8;
    int c = (int)(idx - m * I8) * 8;
    // x / gate columns are interleaved in groups of 32 (same packing as the fused GEMM epilogue)
    const int xc = (c >> 5) * 64 + (c & 31);
    half8 a = ldg_half8(x + m * 2 * I + xc);
    half8 g = ldg_half8(x + m * 2 * I + xc + 32);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] * gelu_f((float)g[j]));
    *reinterpret_cast<half8*>(out + m * I + c) = o;
  }
}
int launch_geglu(const half_t* x, int M, int I, half_t* out, hipStream_t st) {
  if (I & 31) return -3;
  size_t total = (size_t)M * (I >> 3);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  geglu_kernel<<<blocks, 256, 0, st>>>(x, M, I, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ row softmax
// One wavefront per row, in place (VAE AttentionBlock: 4096-wide rows; my_diffusers/models/attention.py:77).
__global__ void __launch_bounds__(256) softmax_rows_kernel(half_t* __restrict__ x, int M, int N, int ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  half_t* p = x + (size_t)row * ld;
  float mx = -INFINITY;
  for (int c = lane * 8; c < N; c += 64 * 8) {
    half8 v = ldg_half8(p + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[j]);
  }
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane * 8; c < N; c += 64 * 8) {
    half8 v = ldg_half8(p + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __expf((float)v[j] - mx);
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane * 8; c < N; c += 64 * 8) {
    half8 v = ldg_half8(p + c);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(__expf((float)v[j] - mx) * inv);
    *reinterpret_cast<half8*>(p + c) = o;
  }
}
int launch_softmax_rows(half_t* x, int M, int N, int ld, hipStream_t st) {
  if ((N & 7) || (ld & 7)) return -3;
  softmax_rows_kernel<<<(M + 3) / 4, 256, 0, st>>>(x, M, N, ld);
  return (int)hipGetLastError();
}

// fp32 variant for the materialised-attention call-back path (attention_control.py:40-41: softmax over the key axis of the
// [B*heads, N, M] score tensor the reference hands to its controller).  One wavefront per row, in place.
__global__ void __launch_bounds__(256) softmax_rows_f32_kernel(float* __restrict__ x, size_t M, int N) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* p = x + row * N;
  float mx = -INFINITY;
  for (int c = lane; c < N; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += __expf(p[c] - mx);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < N; c += 64) p[c] = __expf(p[c] - mx) * inv;
}
int launch_softmax_rows_f32(float* x, size_t M, int N, hipStream_t st) {
  softmax_rows_f32_kernel<<<(unsigned)((M + 3) / 4), 256, 0, st>>>(x, M, N);
  return (int)hipGetLastError();
}
// in [M][N] fp32 -> out [M][ld] fp16, columns [N, ld) zero (the probabilities as the MFMA operand of the P V product)
__global__ void f32_rows_to_f16_padded_kernel(const float* __restrict__ in, size_t M, int N, int ld, half_t* __restrict__ out) {
  const size_t total = M * (size_t)ld;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t m = idx / ld;
    const int n = (int)(idx - m * ld);
    out[idx] = n < N ? (half_t)in[m * N + n] : (half_t)0.f;
  }
}
int launch_f32_rows_to_f16_padded(const float* in, size_t M, int N, int ld, half_t* out, hipStream_t st) {
  size_t total = M * (size_t)ld;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  f32_rows_to_f16_padded_kernel<<<blocks, 256, 0, st>>>(in, M, N, ld, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ GEMV
// out[n] = bias[n] + sum_k act(x[k]) * W[n][k]; one wavefront per output, fp32 accumulate (time-embedding MLP).
__global__ void __launch_bounds__(256) gemv_kernel(const float* __restrict__ x, int K, const half_t* __restrict__ W, int N,
                                                   const float* __restrict__ bias, const float* __restrict__ bias2, int silu_in,
                                                   float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    half8 w = ldg_half8(W + (size_t)n * K + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xv = x[k + j];
      if (silu_in) xv = silu_f(xv);
      acc += xv * (float)w[j];
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) out[n] = acc + (bias ? bias[n] : 0.f) + (bias2 ? bias2[n] : 0.f);
}
int launch_gemv(const float* x, int K, const half_t* W, int N, const float* bias, const float* bias2, int silu_in, float* out,
                hipStream_t st) {
  if (K & 7) return -3;
  gemv_kernel<<<(N + 3) / 4, 256, 0, st>>>(x, K, W, N, bias, bias2, silu_in, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ conversions
__global__ void nchw_f32_to_nhwc_f16_kernel(const float* __restrict__ in, int B, int C, int HW, int Cp, half_t* __restrict__ out) {
  const size_t total = (size_t)B * HW * Cp;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % Cp);
    size_t bp = idx / Cp;
    int pix = (int)(bp % HW);
    int b = (int)(bp / HW);
    out[idx] = c < C ? (half_t)in[((size_t)b * C + c) * HW + pix] : (half_t)0.f;
  }
}
int launch_nchw_f32_to_nhwc_f16(const float* in, int B, int C, int HW, int Cp, half_t* out, hipStream_t st) {
  size_t total = (size_t)B * HW * Cp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  nchw_f32_to_nhwc_f16_kernel<<<blocks, 256, 0, st>>>(in, B, C, HW, Cp, out);
  return (int)hipGetLastError();
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, size_t n, half_t* __restrict__ out) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
    out[idx] = (half_t)in[idx];
}
int launch_f32_to_f16(const float* in, size_t n, half_t* out, hipStream_t st) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  f32_to_f16_kernel<<<blocks, 256, 0, st>>>(in, n, out);
  return (int)hipGetLastError();
}

// utils/utils.py:76-77: image.float() / 127.5 - 1, HWC u8 -> NHWC fp16 with the channel dim zero-padded to Cp.
__global__ void img_u8_to_nhwc_kernel(const uint8_t* __restrict__ img, int n, int HW, int Cp, half_t* __restrict__ out) {
  const size_t total = (size_t)n * HW * Cp;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % Cp);
    size_t bp = idx / Cp;
    float v = 0.f;
    if (c < 3) v = __fsub_rn(__fdiv_rn((float)img[bp * 3 + c], 127.5f), 1.0f);
    out[idx] = (half_t)v;
  }
}
int launch_img_u8_to_nhwc(const uint8_t* img, int n, int HW, int Cp, half_t* out, hipStream_t st) {
  size_t total = (size_t)n * HW * Cp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  img_u8_to_nhwc_kernel<<<blocks, 256, 0, st>>>(img, n, HW, Cp, out);
  return (int)hipGetLastError();
}

// utils/utils.py:62-65: (x / 2 + 0.5).clamp(0, 1) * 255 -> uint8 (truncation), NCHW fp32 -> HWC u8.
__global__ void dec_to_u8_kernel(const float* __restrict__ nchw, int n, int HW, uint8_t* __restrict__ out) {
  const size_t total = (size_t)n * HW * 3;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % 3);
    size_t bp = idx / 3;
    int pix = (int)(bp % HW);
    int b = (int)(bp / HW);
    float v = nchw[((size_t)b * 3 + c) * HW + pix];
    v = __fadd_rn(__fdiv_rn(v, 2.0f), 0.5f);
    v = fminf(fmaxf(v, 0.f), 1.f);
    v = __fmul_rn(v, 255.0f);
    out[idx] = (uint8_t)v;
  }
}
int launch_dec_to_u8(const float* nchw, int n, int HW, uint8_t* out_hwc, hipStream_t st) {
  size_t total = (size_t)n * HW * 3;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  dec_to_u8_kernel<<<blocks, 256, 0, st>>>(nchw, n, HW, out_hwc);
  return (int)hipGetLastError();
}

__global__ void scale_f32_kernel(const float* __restrict__ in, size_t n, float s, float* __restrict__ out) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
    out[idx] = __fmul_rn(in[idx], s);
}
int launch_scale_f32(const float* in, size_t n, float s, float* out, hipStream_t st) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  scale_f32_kernel<<<blocks, 256, 0, st>>>(in, n, s, out);
  return (int)hipGetLastError();
}

__global__ void gather_rows_f32_kernel(const float* __restrict__ in, const int* __restrict__ rows, int nrows, size_t row_elems,
                                       float* __restrict__ out) {
  const size_t total = (size_t)nrows * row_elems;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx / row_elems, e = idx - r * row_elems;
    out[idx] = in[(size_t)rows[r] * row_elems + e];
  }
}
int launch_gather_rows_f32(const float* in, const int* rows, int nrows, size_t row_elems, float* out, hipStream_t st) {
  size_t n = (size_t)nrows * row_elems;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  gather_rows_f32_kernel<<<blocks, 256, 0, st>>>(in, rows, nrows, row_elems, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ weight repack
// src: PyTorch [rows][cols][taps] (conv [out][in][kh*kw] or linear [out][in], taps = 1)
// dst: fp16 [row'][tap * cin_pad + c], row' = row0 + (dh ? (r / dh) * Dp + r % dh : r)   (attention heads padded to Dp)
__device__ __forceinline__ int ilv_row(int r, int half) {
  if (half <= 0) return r;
  return r < half ? (r >> 5) * 64 + (r & 31) : ((r - half) >> 5) * 64 + 32 + ((r - half) & 31);
}
__global__ void repack_matrix_kernel(const void* __restrict__ src, int src_f16, int rows, int cols, int taps, half_t* __restrict__ dst,
                                     int dst_ld, int cin_pad, int row0, int dh, int Dp, int ilv_half) {
  const size_t total = (size_t)rows * cols * taps;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int tap = (int)(idx % taps);
    size_t rc = idx / taps;
    int c = (int)(rc % cols);
    int r = (int)(rc / cols);
    float v = src_f16 ? (float)((const half_t*)src)[idx] : ((const float*)src)[idx];
    int rr = row0 + (dh > 0 ? (r / dh) * Dp + (r % dh) : ilv_row(r, ilv_half));
    dst[(size_t)rr * dst_ld + (size_t)tap * cin_pad + c] = (half_t)v;
  }
}
int launch_repack_matrix(const void* src, int src_f16, int rows, int cols, int taps, half_t* dst, int dst_ld, int cin_pad, int row0,
                         int dh, int Dp, hipStream_t st, int ilv_half) {
  size_t total = (size_t)rows * cols * taps;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  repack_matrix_kernel<<<blocks, 256, 0, st>>>(src, src_f16, rows, cols, taps, dst, dst_ld, cin_pad, row0, dh, Dp, ilv_half);
  return (int)hipGetLastError();
}
__global__ void repack_vec_kernel(const void* __restrict__ src, int src_f16, int n, float* __restrict__ dst, int ilv_half) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    dst[ilv_row(i, ilv_half)] = src_f16 ? (float)((const half_t*)src)[i] : ((const float*)src)[i];
}
int launch_repack_vec(const void* src, int src_f16, int n, float* dst, hipStream_t st, int ilv_half) {
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  repack_vec_kernel<<<blocks, 256, 0, st>>>(src, src_f16, n, dst, ilv_half);
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// CLIP text encoder helpers (transformers CLIPTextEmbeddings; quick_gelu = x * sigmoid(1.702 x))
// ---------------------------------------------------------------------------------------------------------------
__global__ void embed_tokens_kernel(const int* __restrict__ ids, int M, int T, int H, int vocab, const half_t* __restrict__ tok,
                                    const half_t* __restrict__ pos, half_t* __restrict__ out) {
  const size_t total = (size_t)M * H;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / H), c = (int)(i - (size_t)m * H);
    int id = ids[m];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    out[i] = (half_t)((float)tok[(size_t)id * H + c] + (float)pos[(size_t)(m % T) * H + c]);
  }
}
int launch_embed_tokens(const int* ids, int M, int T, int H, int vocab, const half_t* tok_emb, const half_t* pos_emb, half_t* out,
                        hipStream_t st) {
  const size_t total = (size_t)M * H;
  int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  embed_tokens_kernel<<<blocks, 256, 0, st>>>(ids, M, T, H, vocab, tok_emb, pos_emb, out);
  return (int)hipGetLastError();
}
__global__ void quick_gelu_kernel(half_t* __restrict__ x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = (float)x[i];
    x[i] = (half_t)(v / (1.f + __expf(-1.702f * v)));
  }
}
int launch_quick_gelu(half_t* x, size_t n, hipStream_t st) {
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  quick_gelu_kernel<<<blocks, 256, 0, st>>>(x, n);
  return (int)hipGetLastError();
}
__global__ void f16_to_f32_kernel(const half_t* __restrict__ in, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
int launch_f16_to_f32(const half_t* in, size_t n, float* out, hipStream_t st) {
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  f16_to_f32_kernel<<<blocks, 256, 0, st>>>(in, n, out);
  return (int)hipGetLastError();
}

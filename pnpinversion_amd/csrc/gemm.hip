// Implicit-GEMM conv3x3 / conv1x1 / linear on v_mfma_f32_32x32x16_f16 (gfx950).
//
// Restates, as one kernel family, what the reference executes through torch.nn.Conv2d / torch.nn.Linear in
//   models/edict/my_diffusers/models/resnet.py:289,298,346,358 (3x3 convs), :30,48 (upsample conv), :74,95 (stride-2 conv),
//   models/edict/my_diffusers/models/attention.py:125,134 (1x1 proj_in/out), :230-234 (to_q/k/v/out), :303-333 (GEGLU FF).
//
// Tiling: 256 threads = 4 wavefronts (2 x 2); block tile BM x BN x 64; each wave owns (BM/2) x (BN/2) as 32x32 MFMA tiles.
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so that an accumulator register
// group holds 4 consecutive output channels of one pixel (8-byte NHWC stores, vector bias loads).
// global -> registers -> LDS staging (the conv gather needs per-lane predication, which LDS-DMA cannot do), LDS rows
// padded by 16 B so that ds_read_b128 fragment reads are bank-conflict free (stride 144 B = 36 banks).
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#include "ops.h"
#include "igemm_dma.inc"
#include "igemm_pp.inc"

static constexpr int BK = 64;
static constexpr int LDS_LD = BK + 8;  // halfs

// problem z of a batched launch (GemmP::nbatch): the operand / output pointers of that problem
__device__ __forceinline__ void gemm_batch_offsets(GemmP& p, int z) {
  p.x1 += (size_t)z * p.sx1;
  p.w += (size_t)z * p.sw;
  if (p.out) p.out += (size_t)z * p.sout;
  if (p.outT) p.outT = (char*)p.outT + (size_t)z * p.soutT;
}

template <int BM, int BN, bool FASTK>
__global__ void __launch_bounds__(256) igemm_kernel(GemmP p_in) {
  GemmP p = p_in;
  if (p.nbatch > 1) gemm_batch_offsets(p, (int)blockIdx.z);
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  constexpr int AV = BM / 32, WV = BN / 32;  // 16-byte vectors per thread per k-chunk
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half_t* sA = reinterpret_cast<half_t*>(smem_raw);  // [2][BM][LDS_LD]
  half_t* sW = sA + 2 * BM * LDS_LD;                 // [2][BN][LDS_LD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kvec = tid & 7, lrow = tid >> 3;
  const int Cin = p.C1 + p.C2;
  const int HoWo = p.Ho * p.Wo;

  int a_y[AV], a_x[AV], a_b[AV];
  bool a_ok[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    int m = m0 + lrow + 32 * i;
    a_ok[i] = m < p.M;
    int b = m / HoWo;
    int r = m - b * HoWo;
    int yo = r / p.Wo;
    int xo = r - yo * p.Wo;
    a_b[i] = b;
    a_y[i] = yo * p.stride - p.pad;
    a_x[i] = xo * p.stride - p.pad;
  }

  const int nchunks = (p.K + BK - 1) / BK;
  int kc0 = 0, kc1 = nchunks;
  if (p.splitk > 1) {
    kc0 = blockIdx.z * p.kchunks_per_split;
    kc1 = min(nchunks, kc0 + p.kchunks_per_split);
  }

  half8 ra[AV], rw[WV];

  auto load_tiles = [&](int kc) {
    const int k = kc * BK + kvec * 8;
    const bool kok = k < p.K;
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      int n = n0 + lrow + 32 * i;
      rw[i] = (kok && n < p.N) ? ldg_half8(p.w + (size_t)n * p.ldw + k) : zero_half8();
    }
    int tap, c;
    if (FASTK) {
      int kk = kc * BK;
      tap = kk / Cin;
      c = kk - tap * Cin + kvec * 8;
    } else {
      tap = k / Cin;
      c = k - tap * Cin;
    }
    int r = 0, s = 0;
    if (p.ksize == 3) { r = tap / 3; s = tap - 3 * r; }
    const half_t* src; int ld, cc;
    if (c < p.C1) { src = p.x1; ld = p.ldx1; cc = c; } else { src = p.x2; ld = p.ldx2; cc = c - p.C1; }
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      int yi = a_y[i] + r, xi = a_x[i] + s;
      bool ok = a_ok[i] && kok && yi >= 0 && xi >= 0;
      if (p.ups) { ok = ok && yi < 2 * p.H && xi < 2 * p.W; yi >>= 1; xi >>= 1; }
      else { ok = ok && yi < p.H && xi < p.W; }
      ra[i] = ok ? ldg_half8(src + ((size_t)(a_b[i] * p.H + yi) * p.W + xi) * ld + cc) : zero_half8();
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i)
      *reinterpret_cast<half8*>(sA + (size_t)(buf * BM + lrow + 32 * i) * LDS_LD + kvec * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < WV; ++i)
      *reinterpret_cast<half8*>(sW + (size_t)(buf * BN + lrow + 32 * i) * LDS_LD + kvec * 8) = rw[i];
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  if (kc0 < kc1) {
    load_tiles(kc0);
    store_tiles(0);
    __syncthreads();
    for (int kc = kc0; kc < kc1; ++kc) {
      const int buf = (kc - kc0) & 1;
      const bool more = kc + 1 < kc1;
      if (more) load_tiles(kc + 1);
      const half_t* bA = sA + (size_t)(buf * BM + wm0 + (lane & 31)) * LDS_LD + (lane >> 5) * 8;
      const half_t* bW = sW + (size_t)(buf * BN + wn0 + (lane & 31)) * LDS_LD + (lane >> 5) * 8;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        half8 wf[NI], af[MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const half8*>(bW + ni * 32 * LDS_LD + kk * 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const half8*>(bA + mi * 32 * LDS_LD + kk * 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma32(wf[ni], af[mi], acc[mi][ni]);
      }
      if (more) store_tiles(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue: accumulator register r of lane l is (n = tile_n + acc_row(r,l), m = tile_m + (l & 31))
  if (p.epi_lds && p.splitk <= 1) {
    // Coalesced epilogue: the output tile is assembled in LDS (the ring is idle now) and written as full rows, 16 bytes per
    // lane.  The residual tile is staged the same way, so out = fp16(alpha*acc + bias + residual) with a single rounding.
    constexpr int OLD = BN + 8;                         // halfs per staged row
    constexpr int VPR = BN / 8;                         // 16-byte vectors per row
    half_t* sOut = reinterpret_cast<half_t*>(smem_raw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.res) {
      for (int idx = tid; idx < BM * VPR; idx += 256) {
        const int r = idx / VPR, v = idx - r * VPR;
        const int m = m0 + r, n = n0 + v * 8;
        half8 val = (m < p.M && n < p.N) ? ldg_half8(p.res + (size_t)m * p.ldres + n) : zero_half8();
        *reinterpret_cast<half8*>(sOut + r * OLD + v * 8) = val;
      }
      __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int ml = wm0 + mi * 32 + (lane & 31);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn0 + ni * 32 + 8 * g + 4 * (lane >> 5);
          const int n = n0 + nl;
          half4* slot = reinterpret_cast<half4*>(sOut + ml * OLD + nl);
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = acc[mi][ni][4 * g + j] * p.alpha;
            if (p.bias && n + j < p.N) o[j] += p.bias[n + j];
          }
          if (p.res) {
            half4 r4 = *slot;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += (float)r4[j];
          }
          half4 h4 = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
          *slot = h4;
        }
      }
    }
    __syncthreads();
    if (!p.geglu) {
      // thread t owns the fixed 8-channel vector v = t % VPR and walks rows: per-channel sums for the GroupNorm that
      // consumes this tensor fall out of the store loop (fp32 sums of the rounded fp16 values the consumer will read)
      float cs[8], cq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { cs[j] = 0.f; cq[j] = 0.f; }
      for (int idx = tid; idx < BM * VPR; idx += 256) {
        const int r = idx / VPR, v = idx - r * VPR;
        const int m = m0 + r, n = n0 + v * 8;
        if (m < p.M && n < p.N) {
          const half8 val = *reinterpret_cast<const half8*>(sOut + r * OLD + v * 8);
          *reinterpret_cast<half8*>(p.out + (size_t)m * p.ldo + n) = val;
          if (p.stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float f = (float)val[j]; cs[j] += f; cq[j] += f * f; }
          }
        }
      }
      if (p.stats) {
        // lanes sharing a vector index are VPR apart: fold them inside the wave, then across the 4 waves through LDS
        float* sSt = reinterpret_cast<float*>(sOut + BM * OLD);       // [4][BN][2]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int off = VPR; off < 64; off <<= 1) { cs[j] += __shfl_xor(cs[j], off, 64); cq[j] += __shfl_xor(cq[j], off, 64); }
        }
        const int lane_ = tid & 63, wv = tid >> 6;
        if (lane_ < VPR) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            sSt[(wv * BN + lane_ * 8 + j) * 2 + 0] = cs[j];
            sSt[(wv * BN + lane_ * 8 + j) * 2 + 1] = cq[j];
          }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4) { s += sSt[(w4 * BN + tid) * 2]; q += sSt[(w4 * BN + tid) * 2 + 1]; }
          float* dst = p.stats + ((size_t)blockIdx.x * p.N + n0 + tid) * 2;
          dst[0] = s;
          dst[1] = q;
        }
      }
    } else {
      // columns come in [x(32) | gate(32)] groups; the block's BN columns hold BN/2 outputs starting at column n0/2
      constexpr int VPO = BN / 16;
      for (int idx = tid; idx < BM * VPO; idx += 256) {
        const int r = idx / VPO, v = idx - r * VPO;
        const int m = m0 + r;
        const int xc = (v >> 2) * 64 + (v & 3) * 8;     // local column of the x vector; its gate sits 32 columns further
        const int no = (n0 >> 1) + v * 8;
        if (m < p.M && no < (p.N >> 1)) {
          half8 x = *reinterpret_cast<const half8*>(sOut + r * OLD + xc);
          half8 gt = *reinterpret_cast<const half8*>(sOut + r * OLD + xc + 32);
          half8 o8;
#pragma unroll
          for (int j = 0; j < 8; ++j) o8[j] = (half_t)((float)x[j] * gelu_f((float)gt[j]));
          *reinterpret_cast<half8*>(p.out + (size_t)m * p.ldo + no) = o8;
        }
      }
    }
    return;
  }

#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm0 + mi * 32 + (lane & 31);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = n0 + wn0 + ni * 32 + 8 * g + 4 * (lane >> 5);
        float v[4] = {acc[mi][ni][4 * g + 0], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
        if (p.splitk > 1) {
          if (m < p.M) {
            float* dst = p.slab + ((size_t)blockIdx.z * p.M + m) * p.N + nb;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (nb + j < p.N) dst[j] = v[j];
          }
        } else {
          epilogue_store4(p, m, nb, v, p.bias);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// v2: asynchronous LDS-DMA staging (global_load_lds_dwordx4), no staging registers, no ds_write pass.
//   * each wave-level DMA instruction fills 1 KiB of LDS lane-linearly, so the bank-conflict-free layout is obtained by
//     permuting the per-lane SOURCE address and applying the same involution on the fragment read:
//       logical (row R, 16-byte chunk s)  ->  byte  (R>>4)*2048 + (R&7)*256 + ((R>>3)&1)*128 + ((s ^ (R&7)) * 16)
//     (a ds_read_b128 lane group then touches 16 distinct 16-byte slots of the 256-byte bank row);
//   * conv zero padding / M,N tails: out-of-range lanes read from a zeroed page instead of being masked (a masked DMA lane
//     would leave stale LDS bytes);
//   * two LDS stages (64 KiB at 128x128 -> two blocks per CU); the DMA of chunk k+1 flies under the MFMAs of chunk k and is
//     drained by the vmcnt(0) the compiler puts in front of the one barrier per chunk.
// Requires (C1 + C2) % 64 == 0 and C1 % 64 == 0 (every SD-1.x layer but the 4/3-channel stems, which stay on v1).
// ------------------------------------------------------------------------------------------------------------------
static int g_tile_order = -1;   // PNPI_TILE_ORDER: force 0 / 1 (ablation)

static thread_local int g_last_geom[7] = {0, 0, 0, 0, 0, 0, 0};   // template arguments of the most recent igemm_dma_kernel launch (all 0: the v1 kernel)
template <int BM, int BN, int BKT, int NST, int WGM = 2, int ABL = 0, int WK = 1>
static int launch_dma(const GemmP& p_in, dim3 grid, hipStream_t st, const half_t* zero_page) {
  using GEO = IgemmGeom<BM, BN, BKT, NST, WGM, ABL, WK>;
  g_last_geom[0] = BM; g_last_geom[1] = BN; g_last_geom[2] = BKT; g_last_geom[3] = NST; g_last_geom[4] = WGM; g_last_geom[5] = ABL; g_last_geom[6] = WK;
  GemmP p = p_in;
  p.gx = grid.x; p.gy = grid.y; p.gz = grid.z;
  {
    const double bytes_a = 2.0 * p.B * p.H * p.W * (p.C1 + p.C2), bytes_w = 2.0 * (double)p.N * p.K;
    p.tile_order = (bytes_a + 8 * bytes_w <= 8 * bytes_a + bytes_w) ? 0 : 1;
    if (g_tile_order >= 0) p.tile_order = g_tile_order;
  }
  const int total = (int)(grid.x * grid.y * grid.z);
  grid = dim3((unsigned)(((total + 7) / 8) * 8), 1, 1);
  constexpr int lds = GEO::LDS;
  static DeviceOnce attr_once;
  if (int r = once_per_device(attr_once, [&]() { return (int)hipFuncSetAttribute((const void*)igemm_dma_kernel<BM, BN, BKT, NST, WGM, ABL, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); })) return r;
  igemm_dma_kernel<BM, BN, BKT, NST, WGM, ABL, WK><<<grid, GEO::NT_ALL, lds, st>>>(p, zero_page);
  return 0;
}

// The 8-wave ping-pong kernel (igemm_pp.inc): one workgroup per CU, same tile-order / grid conventions as launch_dma.
template <int BM, int BN, int MI0, int NI0, int ABL = 0>
static int launch_pp(const GemmP& p_in, dim3 grid, hipStream_t st, const half_t* zero_page) {
  using GEO = PpGeom<BM, BN, MI0, NI0>;
  g_last_geom[0] = BM; g_last_geom[1] = BN; g_last_geom[2] = MI0; g_last_geom[3] = NI0; g_last_geom[4] = ABL; g_last_geom[5] = 0; g_last_geom[6] = -1;   // [6] = -1: igemm_pp_kernel
  GemmP p = p_in;
  p.gx = grid.x; p.gy = grid.y; p.gz = grid.z;
  {
    const double bytes_a = 2.0 * p.B * p.H * p.W * (p.C1 + p.C2), bytes_w = 2.0 * (double)p.N * p.K;
    p.tile_order = (bytes_a + 8 * bytes_w <= 8 * bytes_a + bytes_w) ? 0 : 1;
    if (g_tile_order >= 0) p.tile_order = g_tile_order;
  }
  const int total = (int)(grid.x * grid.y * grid.z);
  grid = dim3((unsigned)(((total + 7) / 8) * 8), 1, 1);
  constexpr int lds = GEO::LDS;
  static DeviceOnce attr_once;
  if (int r = once_per_device(attr_once, [&]() { return (int)hipFuncSetAttribute((const void*)igemm_pp_kernel<BM, BN, MI0, NI0, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); })) return r;
  (void)zero_page;      // the ping-pong kernel's out-of-range lanes read zeros through the buffer descriptor's bounds check (igemm_pp.inc)
  igemm_pp_kernel<BM, BN, MI0, NI0, ABL><<<grid, GEO::NT, lds, st>>>(p);
  return 0;
}

// Deterministic split-K combine: fixed slab order, then the common epilogue.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmP p) {
  const int groups_per_row = (p.N + 3) / 4;
  const size_t total = (size_t)p.M * groups_per_row;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    int m = (int)(idx / groups_per_row);
    int nb = (int)(idx - (size_t)m * groups_per_row) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splitk; ++z) {
      const float* src = p.slab + ((size_t)z * p.M + m) * p.N + nb;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (nb + j < p.N) v[j] += src[j];
    }
    epilogue_store4(p, m, nb, v, p.bias);
  }
}

// The same combine for the layouts every layer of the UNet has (N % 4 == 0, plain row-major output, 16-byte aligned bias, 8-byte
// aligned rows): one thread per 4 columns, unconditional 16-byte slab loads unrolled over the slabs (the generic kernel's
// predicated scalar loads each carry their own wait), same summation order -> same bits.
__global__ void __launch_bounds__(256) splitk_reduce_vec_kernel(GemmP p) {
  const int gpr = p.N >> 2;
  const size_t total = (size_t)p.M * gpr;
  const size_t slab_stride = (size_t)p.M * p.N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / gpr);
    const int nb = (int)(idx - (size_t)m * gpr) * 4;
    const float* src = p.slab + (size_t)m * p.N + nb;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int z = 0; z < p.splitk; ++z) v += *reinterpret_cast<const floatx4*>(src + (size_t)z * slab_stride);
    v *= p.alpha;
    if (p.bias) v += *reinterpret_cast<const floatx4*>(p.bias + nb);
    if (p.res) {
      const half4 r4 = *reinterpret_cast<const half4*>(p.res + (size_t)m * p.ldres + nb);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += (float)r4[j];
    }
    half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4*>(p.out + (size_t)m * p.ldo + nb) = h;
  }
}

void gemm_defaults(GemmP& p) {
  p.x1 = nullptr; p.x2 = nullptr; p.C1 = 0; p.C2 = 0; p.ldx1 = 0; p.ldx2 = 0;
  p.B = 1; p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.ksize = 1; p.stride = 1; p.pad = 0; p.ups = 0;
  p.w = nullptr; p.ldw = 0; p.M = 0; p.N = 0; p.K = 0; p.bias = nullptr; p.res = nullptr; p.ldres = 0; p.alpha = 1.f;
  p.out = nullptr; p.ldo = 0; p.outT = nullptr; p.vt_col0 = 1 << 30; p.vt_ld = 0; p.vt_f32 = 0; p.rows_per_batch = 1; p.vt_perm16 = 0;
  p.slab = nullptr; p.splitk = 1; p.kchunks_per_split = 0; p.geglu = 0; p.epi_lds = 0; p.stats = nullptr; p.res_late = 0; p.bias_init = 0;
  p.k_order = 0; p.nbatch = 1; p.sx1 = 0; p.sw = 0; p.sout = 0; p.soutT = 0;
}

static constexpr size_t lds_bytes(int BM, int BN) { return (size_t)(2 * BM + 2 * BN) * LDS_LD * sizeof(half_t); }

static half_t* g_zero_page = nullptr;
static constexpr size_t ZERO_PAGE_BYTES = 128 << 10;   // >= 2 * (longest K + one chunk): out-of-range rows walk it like real rows
static int g_wide = 1;      // PNPI_IGEMM_WIDE=0: never pick the 128x320 / 128x256 tiles (ablation)
static int g_use_dma = 1;      // 0: register-staged v1 kernel everywhere
static int g_use_table = 1;    // tuning "igemm_table" = 0: cost model only (no measured per-shape table)
static int g_table_near = 1;   // tuning "igemm_table_near" = 0: exact {M, N, K, ksize} matches only (no nearest-row-count entry)
static int g_res_late = 0;     // tuning "igemm_res_late" = 1: residual added in the store loop (fp16(fp16(acc + bias) + res)) instead of staged
static int g_vt_lds = 1;       // tuning "igemm_vt_lds" = 0: transposed columns through the scalar epilogue (A/B)
static int g_deep_rings = 1;   // tuning "igemm_deep_rings" = 0: shallow rings whatever the occupancy (A/B)
static int g_bias_init = 1;     // 0: bias added in the epilogue (ablation)
static int g_pp_only_n = 0, g_pp_only_k = 0, g_pp_only_m = 0;   // tuning "igemm_pp_only_{m,n,k}" > 0: table entries of the ping-pong kernel apply only to launches with that M / N / K (bisection of a wrong layer)
static int g_tapin = 0;         // tuning "igemm_tapin" = 1: the ping-pong kernel walks 3 x 3 convolutions channel-slab-major (GemmP::k_order)
static int g_varpp = 0;         // tuning "igemm_vpp": ablations of the ping-pong kernel (1 = no MFMAs, 2 = no DMA; both produce garbage)
static int g_sched = 0;         // tuning "igemm_sched" = 1: the hand-scheduled main loop (igemm_dma_kernel<..., ABL = 4>) for the one-k-group tile configurations
static int g_force_split = 0;   // > 0 with igemm_force_cfg: split-K of every auto-configured launch (in-forward tuning sweeps)
static thread_local int g_last_cfg = -1, g_last_split = 1;   // per thread: two contexts may launch from two threads   // what the most recent launch_igemm used (profiling dumps)
void igemm_last_launch(int* cfg, int* split, int* geom) { *cfg = g_last_cfg; *split = g_last_split; for (int i = 0; i < 7; ++i) geom[i] = g_last_geom[i]; }
static int g_force_cfg = -1;   // >= 0: every auto-configured launch uses this tile configuration (tests, whole-forward A/B)
static int g_var128 = 2, g_var64 = 0, g_var256 = 0, g_var320 = 1, g_var256n = 1;
static long g_v128_bk64_tiles = 0;   // PNPI_V128_BK64_TILES: tile count from which the 128x128 kernel switches to 128-byte rows   // tuning variants (PNPI_IGEMM_V128 / PNPI_IGEMM_V64)
void igemm_set_dma(int on) { g_use_dma = on; }
// process-wide tuning knobs (A/B measurements inside one process, tests of the non-default variants); 0 on success
int igemm_set_tuning(const char* key, int v) {
  struct { const char* k; int* p; } tab[] = {{"igemm_dma", &g_use_dma}, {"igemm_v128", &g_var128}, {"igemm_v64", &g_var64}, {"igemm_v256", &g_var256},
                                             {"igemm_v320", &g_var320}, {"igemm_v256n", &g_var256n}, {"igemm_wide", &g_wide}, {"tile_order", &g_tile_order}, {"igemm_force_cfg", &g_force_cfg}, {"igemm_force_split", &g_force_split}, {"igemm_bias_init", &g_bias_init}, {"igemm_sched", &g_sched}, {"igemm_vpp", &g_varpp}, {"igemm_tapin", &g_tapin}, {"igemm_pp_only_n", &g_pp_only_n}, {"igemm_pp_only_k", &g_pp_only_k}, {"igemm_pp_only_m", &g_pp_only_m}, {"igemm_deep_rings", &g_deep_rings}, {"igemm_vt_lds", &g_vt_lds}, {"igemm_res_late", &g_res_late}, {"igemm_table", &g_use_table}, {"igemm_table_near", &g_table_near}};
#ifndef PNPI_ABLATIONS
  // the ablation instances (no MFMAs / no DMA / no fragment reads, the hand-scheduled 4-wave loop) exist only in a `build --ablations`
  // library: a product build rejects their selectors here instead of silently timing the default kernel
  if ((!strcmp(key, "igemm_sched") || !strcmp(key, "igemm_vpp")) && v != 0) return -2;
  if ((!strcmp(key, "igemm_v128") || !strcmp(key, "igemm_v320")) && (v == 11 || v == 12 || v == 15)) return -2;
#endif
  for (auto& e : tab)
    if (!strcmp(key, e.k)) { *e.p = v; return 0; }
  return -1;
}

int igemm_init() {
  if (const char* e = getenv("PNPI_IGEMM_DMA")) g_use_dma = atoi(e);
  if (const char* e = getenv("PNPI_IGEMM_V128")) g_var128 = atoi(e);
  if (const char* e = getenv("PNPI_IGEMM_V64")) g_var64 = atoi(e);
  if (const char* e = getenv("PNPI_IGEMM_V256")) g_var256 = atoi(e);
  if (const char* e = getenv("PNPI_IGEMM_V320")) g_var320 = atoi(e);
  if (const char* e = getenv("PNPI_IGEMM_V256N")) g_var256n = atoi(e);
  if (const char* e = getenv("PNPI_IGEMM_WIDE")) g_wide = atoi(e);
  if (const char* e = getenv("PNPI_V128_BK64_TILES")) g_v128_bk64_tiles = atol(e);
  if (const char* e = getenv("PNPI_TILE_ORDER")) g_tile_order = atoi(e);
  if (!g_zero_page) {
    HIP_CHECK_RET(hipMalloc((void**)&g_zero_page, ZERO_PAGE_BYTES));
    HIP_CHECK_RET(hipMemset(g_zero_page, 0, ZERO_PAGE_BYTES));
  }
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(128, 128)));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(128, 128)));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<64, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(64, 64)));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<64, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(64, 64)));
  return 0;
}

// Tile configurations of the LDS-DMA kernel (cfg ids of launch_igemm's force_cfg / tools): 0 = 128x128, 1 = 64x64 (2 = the same
// with split-K forced), 3 = 256x128 (ablation), 4 = 128x320, 5 = 128x256.
struct TileCfg { int id, bm, bn; double rate, t_fix; int bpc; };   // rate: FLOP/s of the whole chip with every CU full; t_fix: per-tile
static const TileCfg kTiles[] = {                                  // prologue + epilogue seconds; bpc: co-resident blocks per CU
  {0, 128, 128, 868e12, 1.22e-6, 3},
  {1, 64, 64, 572e12, 0.25e-6, 4},
  {4, 128, 320, 1228e12, 5.15e-6, 2},
  {5, 128, 256, 1178e12, 3.61e-6, 2},
};
// Measured choices for the SD-1.x layer shapes at the row counts of the benchmarked schedule (1-row inversion, 12-row lock step):
// per-shape best of every tile / split-K / k-group configuration (tools/autotune2.py -> tools/gen_tile_table.py).  The cost model
// below covers every other shape (other row counts, other model widths).
struct TileEntry { int M, N, K, ks, cfg, split; };
static const TileEntry kTileTable[] = {
#include "tile_table.inc"
};
// Table lookup: the exact {M, N, K, ksize}; else -- the same layer (N, K, ksize) at another row count, e.g. --batch_size 2 ... 7 of the
// sweep driver -- the entry whose M is nearest in ratio, up to 4x away.  Held out of the table one row count at a time, the nearest
// entry's configuration costs 13.3 / 10.1 / 6.7 / 5.2 ms per 12- / 8- / 4- / 3-row forward against 15.3 / 10.4 / 6.9 / 6.1 ms for the
// cost model's pick (per-shape best 12.5 / 9.4 / 6.0 / 5.1; profiles/round2_fwd_tune_b*.json).  how: 1 exact, 2 nearest, 0 none.
static const TileEntry* tile_table_lookup(int M, int N, int K, int ks, int* how) {
  const TileEntry* near = nullptr;
  double near_ratio = 4.0 + 1e-9;
  for (const TileEntry& e : kTileTable) {
    if (e.N != N || e.K != K || e.ks != ks) continue;
    if (e.M == M) { if (how) *how = 1; return &e; }
    const double r = e.M > M ? (double)e.M / M : (double)M / e.M;
    if (r < near_ratio) { near_ratio = r; near = &e; }      // ties: the first (smaller M) entry of the sorted table
  }
  if (!g_table_near) near = nullptr;
  if (how) *how = near ? 2 : 0;
  return near;
}
int igemm_table_lookup(int M, int N, int K, int ks, int* cfg, int* split, int* entry_m) {
  int how = 0;
  const TileEntry* e = (M > 0 && N > 0 && K > 0) ? tile_table_lookup(M, N, K, ks, &how) : nullptr;
  if (e) { if (cfg) *cfg = e->cfg; if (split) *split = e->split; if (entry_m) *entry_m = e->M; }
  return how;
}
// cfg 6 = 256x320, 7 = 256x256 (8 waves, 128-byte rows, two stages, one block per CU); 8-11: K-parallel wave groups; 12 = 64x320;
// 13 = 128x128 with a two-stage ring; 14 / 15 = 128x128 / 128x256 with 128-byte rows; 18 = 128x128, 128-byte rows, three stages:
// reached through the table or force_cfg

// product tile configurations with one k-group: compiler-scheduled main loop, or (tuning "igemm_sched") the hand-scheduled one
#ifdef PNPI_ABLATIONS
#define LDMA(...) (g_sched == 1 ? launch_dma<__VA_ARGS__, 2, 4>(p, grid, st, g_zero_page) : g_sched == 2 ? launch_dma<__VA_ARGS__, 2, 5>(p, grid, st, g_zero_page) : launch_dma<__VA_ARGS__>(p, grid, st, g_zero_page))
#else
#define LDMA(...) launch_dma<__VA_ARGS__>(p, grid, st, g_zero_page)
#endif
#define LDMA0(...) launch_dma<__VA_ARGS__>(p, grid, st, g_zero_page)
// Can the 8-wave ping-pong kernel (cfg 16 = 256 x 256, 17 = 192 x 320) take this launch?  It has no scalar epilogue: a launch either takes
// the LDS epilogue (whole tiles plain or transposed, 16-byte rows, 16-byte aligned bias) or writes split-K slabs; its DMA addresses are
// 32-bit byte offsets under a 2 GiB buffer descriptor (operands beyond that stay on the 64-bit-pointer kernels); batched launches stay
// on the 4-wave kernels.
static bool pp_eligible(const GemmP& p, int cfg, int split, bool dma_ok) {
  const int bn_pp = cfg == 17 ? 320 : 256;
  const bool vt_none_ = p.vt_col0 >= p.N;
  const bool vt_lds_ = !vt_none_ && p.outT && !p.vt_f32 && p.vt_col0 % bn_pp == 0 && p.rows_per_batch % 8 == 0 && p.vt_ld % 8 == 0 && p.M % 8 == 0 && !p.res &&
                       !p.geglu && g_vt_lds;
  const bool epi_ok = (vt_none_ || vt_lds_) && (p.N % 8 == 0) && (p.vt_col0 == 0 || p.ldo % 8 == 0) && (!p.res || p.ldres % 8 == 0) &&
                      (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.vt_perm16 || p.rows_per_batch % 16 == 0);
  const double lim = 2147483648.0 - 1048576.0;
  const bool off32_ok = 2.0 * p.B * p.H * p.W * (double)(p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) < lim && 2.0 * (double)p.N * p.ldw < lim;
  return dma_ok && off32_ok && !(split == 1 && !epi_ok) && !(split > 1 && p.N % 4 != 0) && p.nbatch <= 1;
}

int launch_igemm(GemmP p, float* ws, size_t ws_bytes, hipStream_t st, int force_cfg, int force_split, int* cfg_used,
                 int* stats_tile_rows, GemmP* deferred) {
  if (deferred) deferred->splitk = 1;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -2;
  const int Cin = p.C1 + p.C2;
  if ((Cin & 7) || (p.K & 7) || (p.C1 & 7) || (p.ldw & 7) || (p.ldx1 & 7) || (p.C2 && (p.ldx2 & 7))) return -3;
  if (p.K != p.ksize * p.ksize * Cin) return -4;
  if (p.vt_col0 > p.N) p.vt_col0 = p.N;
  const bool fast = (Cin % BK == 0) && (p.C1 % BK == 0);
  const int nchunks = (p.K + BK - 1) / BK;
  const bool dma_ok = fast && g_use_dma && (p.K % BK == 0) && (p.ldw % 8 == 0) && ((size_t)p.K * 2 + 1024 <= ZERO_PAGE_BYTES);
  auto tiles_of = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const long t64 = tiles_of(64, 64);
  int cfg = force_cfg;
  int split = 1;
  if (cfg < 0 && g_force_cfg >= 0) { cfg = g_force_cfg == 2 ? 1 : g_force_cfg; if (force_split <= 0) force_split = g_force_split; }
  if (cfg < 0 && g_use_table && dma_ok) {
    if (const TileEntry* pe = tile_table_lookup(p.M, p.N, p.K, p.ksize, nullptr)) {
        const TileEntry& e = *pe;
        const int ebn = (e.cfg == 4 || e.cfg == 6 || e.cfg == 12 || e.cfg == 17) ? 320 : ((e.cfg == 5 || e.cfg == 7 || e.cfg == 15 || e.cfg == 16) ? 256 : ((e.cfg == 1 || e.cfg == 8 || e.cfg == 11) ? 64 : 128));
        const bool split_ok = e.split == 1 || (!p.geglu && ws && (size_t)e.split * p.M * p.N * sizeof(float) <= ws_bytes);
        const bool vt_ok = p.vt_col0 >= p.N || p.vt_col0 % ebn == 0;
        const bool pp_masked = (e.cfg == 16 || e.cfg == 17) && ((g_pp_only_n > 0 && p.N != g_pp_only_n) || (g_pp_only_k > 0 && p.K != g_pp_only_k) || (g_pp_only_m > 0 && p.M != g_pp_only_m) ||
                                                                 g_pp_only_n < 0);
        // a ping-pong entry (tile and split tuned together) applies only to a launch that kernel can take: otherwise the cost model
        // picks tile AND split for the 4-wave kernels (not the entry's split on a 128 x 128 tile)
        const bool pp_ok = !(e.cfg == 16 || e.cfg == 17) || pp_eligible(p, e.cfg, e.split, dma_ok);
        if (split_ok && vt_ok && !pp_masked && pp_ok && (g_wide || e.cfg < 4 || e.cfg == 13 || e.cfg == 14 || e.cfg == 18)) { cfg = e.cfg; split = e.split; }
    }
  }
  if (cfg < 0) {
    // Tile / split-K selection by a small cost model (constants fitted to per-shape timings on MI355X, tools/fit_cost_model.py):
    //   time ~ (units on the busiest CU) x (padded FLOPs of one unit) / (per-CU rate of the tile x co-residency factor)
    //          + split-K slab traffic + the reduce launch.
    // Wider tiles move fewer operand bytes per FLOP through the LDS-DMA path (the limiter), but need split-K on the
    // low-resolution layers to put work on all 256 CUs.
    double best = 1e30;
    static const int splits[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (const TileCfg& tc : kTiles) {
      if (tc.id >= 4 && !(dma_ok && g_wide)) continue;     // the wide tiles exist only as LDS-DMA kernels
      const long tiles = tiles_of(tc.bm, tc.bn);
      for (int s : splits) {
        if (s > 1 && (nchunks / s < 4 || (size_t)s * p.M * p.N * sizeof(float) > ws_bytes || ws == nullptr)) continue;
        if (s > 1 && p.geglu) continue;
        // constants: tools/fit_cost_model2.py on tools/autotune2.py timings of every configuration per layer shape (rms log error
        // 0.10; the picks cost 1 % more than the per-shape best over a 12-row forward)
        const double unit = tc.t_fix + (double)((nchunks + s - 1) / s) * (2.0 * tc.bm * tc.bn * BK) / (tc.rate / 256.0);   // padded tiles do real work
        const long units = tiles * s;
        const long on_busiest = (units + 255) / 256;
        const double per_cu = (double)units / 256.0;
        // co-resident blocks cover each other's exposed loads: a lone block on a CU runs below the tile's full rate
        const double fill = per_cu >= tc.bpc ? 1.0 : (per_cu <= 1.0 ? 0.0 : (per_cu - 1.0) / (tc.bpc - 1.0));
        const double resid = 0.82 + 0.18 * fill;
        double t = (double)on_busiest * unit / resid;
        if (s > 1) t += (double)(2 * s + 1) * p.M * p.N * 4.0 / 8.14e12 + 4.97e-6;   // slabs written, re-read, output + the reduce launch
        if (p.vt_col0 < p.N && p.vt_col0 % tc.bn != 0) t *= 1.3;   // transposed columns not tile-aligned: scalar epilogue
        if (t < best) { best = t; cfg = tc.id; split = s; }
      }
    }
  } else if (cfg == 2) {
    cfg = 1;
    split = force_split > 0 ? force_split : (int)((512 + t64 - 1) / t64);
    if (split > 16) split = 16;
    if (split > nchunks) split = nchunks;
    while (split > 1 && nchunks / split < 4) --split;
  } else if (force_split > 1) {
    split = force_split;
  }
  if (p.nbatch > 1) { split = 1; if (cfg >= 8 && cfg <= 11) cfg = (cfg == 8 || cfg == 11) ? 1 : 0; }   // no split-K / k-groups across problems
  if (cfg >= 8 && !dma_ok) cfg = (cfg == 8 || cfg == 11) ? 1 : 0;
  if (cfg >= 3 && !dma_ok) cfg = 0;                                 // 256x128 / 128x320 / 128x256 exist only as LDS-DMA kernels
  // whatever chose the split (cost model or a caller-forced value): the slabs must fit the workspace and each split needs work
  if (split > 1) {
    if (split > nchunks) split = nchunks;
    if (ws == nullptr || (size_t)split * p.M * p.N * sizeof(float) > ws_bytes || p.geglu || cfg == 3) split = 1;
  }
  if ((cfg == 16 || cfg == 17) && !pp_eligible(p, cfg, split, dma_ok)) cfg = 0;     // forced configurations (tests, sweeps): the 128 x 128 kernel instead
  const bool c64 = cfg == 1 || cfg == 8 || cfg == 11 || cfg == 12, c256m = cfg == 3 || cfg == 6 || cfg == 7 || cfg == 16;
  const bool cpp = cfg == 16 || cfg == 17;      // the 8-wave ping-pong kernel: 16 = 256 x 256, 17 = 192 x 320
  if (cfg_used) *cfg_used = split > 1 ? 2 : (((cfg >= 4 && cfg <= 7) || cfg == 12 || (cfg >= 15 && cfg <= 17)) ? 9 : (c64 ? 1 : 0));
  p.splitk = split;
  p.kchunks_per_split = (nchunks + split - 1) / split;
  g_last_cfg = cfg; g_last_split = split;
  for (int i = 0; i < 7; ++i) g_last_geom[i] = 0;
  const int bn_sel = (cfg == 4 || cfg == 6 || cfg == 12 || cfg == 17) ? 320 : ((cfg == 5 || cfg == 7 || cfg == 15 || cfg == 16) ? 256 : (c64 ? 64 : 128));
  const bool vt_none = p.vt_col0 >= p.N;
  // transposed (V^T) columns through the LDS epilogue too, when whole tiles are either plain or transposed and 8-token runs stay
  // inside one batch item
  const bool vt_lds = !vt_none && dma_ok && p.outT && !p.vt_f32 && p.vt_col0 % bn_sel == 0 && p.rows_per_batch % 8 == 0 && p.vt_ld % 8 == 0 &&
                      p.M % 8 == 0 && !p.res && !p.geglu && g_vt_lds;
  p.epi_lds = (vt_none || vt_lds) && (p.N % 8 == 0) && (p.vt_col0 == 0 || p.ldo % 8 == 0) && (!p.res || p.ldres % 8 == 0) && split == 1;
  if (vt_lds) p.stats = nullptr;
  p.res_late = g_res_late;
  p.bias_init = (dma_ok && split == 1 && p.bias && p.alpha == 1.f && p.N % 4 == 0 && ((uintptr_t)p.bias & 15) == 0 && g_bias_init && !cpp) ? 1 : 0;
  if (p.vt_perm16 && (p.rows_per_batch % 16 != 0 || p.vt_f32)) return -9;   // permuted V^T: whole 16-token groups, fp16
  if (p.geglu && !(p.epi_lds && dma_ok && p.N % 64 == 0)) return -7;   // GEGLU exists only in the LDS epilogue
  if (!(p.epi_lds && dma_ok && !p.geglu)) p.stats = nullptr;       // statistics come only from the DMA kernel's LDS epilogue
  const int bm = cfg == 17 ? 192 : (c256m ? 256 : (c64 ? 64 : 128));
  const int bn = bn_sel;
  if (stats_tile_rows) *stats_tile_rows = p.stats ? (cpp ? 64 : bm) : 0;     // the ping-pong kernel writes its partials per 64-row sub-block
  p.slab = ws;
  if (p.nbatch > 1) {
    if (split != 1 || p.bias || p.res || p.stats || p.geglu || p.x2) return -8;     // batched launches: plain products only
  }
  dim3 grid((p.M + bm - 1) / bm, (p.N + bn - 1) / bn, p.nbatch > 1 ? p.nbatch : split);
  int r = 0;
  // Ring depth by occupancy: with at most one block per CU nothing else covers the HBM latency of the weight stream (cold in a
  // forward: 1.7 GB of weights pass through per UNet call), and the whole 160 KB of LDS is free -- so sparse launches take an
  // 8-deep ring (7 chunks in flight per block); two blocks per CU a 4-deep one; fuller launches the shallow rings that fit 3 blocks.
  const long units = (long)grid.x * grid.y * grid.z;
  const int sparse = !g_deep_rings ? 0 : (units <= 256 ? 2 : (units <= 512 ? 1 : 0));
  if (!dma_ok) {
    if (cfg == 1) {
      if (fast) igemm_kernel<64, 64, true><<<grid, 256, lds_bytes(64, 64), st>>>(p);
      else igemm_kernel<64, 64, false><<<grid, 256, lds_bytes(64, 64), st>>>(p);
    } else {
      if (fast) igemm_kernel<128, 128, true><<<grid, 256, lds_bytes(128, 128), st>>>(p);
      else igemm_kernel<128, 128, false><<<grid, 256, lds_bytes(128, 128), st>>>(p);
    }
  } else if (cfg == 3) {
    r = g_var256 == 1 ? launch_dma<256, 128, 32, 2, 2>(p, grid, st, g_zero_page) : launch_dma<256, 128, 32, 3, 2>(p, grid, st, g_zero_page);
  } else if (cfg == 4) {
    if (sparse == 2 && g_var320 == 1) r = LDMA0(128, 320, 32, 5);
    else switch (g_var320) {
      case 0: r = LDMA0(128, 320, 32, 3); break;       // 84 KB ring = whole-tile epilogue, 1 block / CU
      case 2: r = LDMA0(128, 320, 64, 2); break;       // 128-byte rows, 112 KB, 1 block / CU
#ifdef PNPI_ABLATIONS
      case 11: r = launch_dma<128, 320, 32, 2, 2, 1>(p, grid, st, g_zero_page); break;   // ablation: DMA only
      case 12: r = launch_dma<128, 320, 32, 2, 2, 2>(p, grid, st, g_zero_page); break;   // ablation: compute only
#endif
      default: r = LDMA(128, 320, 32, 2); break;      // 56 KB ring, two-pass epilogue, 2 blocks / CU
    }
  } else if (cfg == 12) {
    r = LDMA0(64, 320, 32, 2);             // 64 x 320: 768 tiles on the 12-row 64 x 64 level = 3 per CU
  } else if (cfg == 13) {
    r = LDMA(128, 128, 32, 2);            // 32 KB ring (two-pass epilogue): 4 blocks / CU for the short-K layers
  } else if (cfg == 18) {
    r = LDMA0(128, 128, 64, 3);           // 128-byte rows, three stages (96 KB, 1 block / CU): the one-wave launches of the 16 x 16 level
  } else if (cfg == 14) {
    r = LDMA(128, 128, 64, 2);            // 128-byte rows, half the barriers per k: 64 KB, 2 blocks / CU
  } else if (cfg == 15) {
    r = LDMA0(128, 256, 64, 2);            // the same for the 128 x 256 tile: 96 KB, 1 block / CU
  } else if (cfg == 8) {
    r = launch_dma<64, 64, 64, 2, 2, 0, 4>(p, grid, st, g_zero_page);      // 16 waves: 4 k-groups
  } else if (cfg == 11) {
    r = launch_dma<64, 64, 64, 2, 2, 0, 2>(p, grid, st, g_zero_page);      // 8 waves: 2 k-groups, 64 KB (2 blocks / CU)
  } else if (cfg == 9) {
    r = launch_dma<128, 128, 32, 3, 2, 0, 2>(p, grid, st, g_zero_page);    // 8 waves: 2 k-groups, 96 KB
  } else if (cfg == 10) {
    r = launch_dma<128, 128, 64, 2, 2, 0, 2>(p, grid, st, g_zero_page);    // 8 waves: 2 k-groups, 128-byte rows, 128 KB
  } else if (cfg == 16 || cfg == 17) {
    p.k_order = (g_tapin && p.ksize == 3) ? 1 : 0;
#define LPP(...) (cfg == 16 ? launch_pp<256, 256, 2, 4, __VA_ARGS__>(p, grid, st, g_zero_page) : launch_pp<192, 320, 1, 4, __VA_ARGS__>(p, grid, st, g_zero_page))
#ifdef PNPI_ABLATIONS      // `python -m pnpinversion_amd.build --ablations`: tools/pp_ablate.py, tools/profile_pp.sh
    switch (g_varpp) {
      case 1: r = LPP(1); break;     // ablations: no MFMAs
      case 2: r = LPP(2); break;     // no DMA
      case 3: r = LPP(3); break;     // DMA only
      case 4: r = LPP(4); break;     // MFMAs only
      default: r = LPP(0); break;
    }
#else
    if (g_varpp) return -10;         // the ablation instances are not in this build
    r = LPP(0);
#endif
#undef LPP
  } else if (cfg == 6) {
    r = launch_dma<256, 320, 64, 2, 4>(p, grid, st, g_zero_page);
  } else if (cfg == 7) {
    r = launch_dma<256, 256, 64, 2, 4>(p, grid, st, g_zero_page);
  } else if (cfg == 5) {
    if (sparse == 2 && g_var256n == 1) r = LDMA0(128, 256, 32, 6);
    else switch (g_var256n) {
      case 0: r = LDMA0(128, 256, 32, 3); break;       // 72 KB, 2 blocks / CU
      case 2: r = LDMA0(128, 256, 64, 2); break;
      default: r = LDMA(128, 256, 32, 2); break;      // 48 KB: two-pass epilogue, 3 blocks / CU by LDS
    }
  } else if (cfg == 0) {
    int var = g_var128;
    // 128-byte rows with 2 stages (64 KB, 2 blocks / CU) beat 64-byte rows with 3 stages (48 KB, 3 blocks / CU) once every CU
    // holds two blocks that cover for each other's exposed loads; below that the deeper ring wins
    if (g_v128_bk64_tiles > 0 && var == 2 && (long)grid.x * grid.y * grid.z >= g_v128_bk64_tiles) var = 0;
    if (var == 2 && sparse == 2) var = 8;
    else if (var == 2 && sparse == 1) var = 3;
    switch (var) {
      case 1: r = LDMA0(128, 128, 64, 3); break;
      case 2: r = LDMA(128, 128, 32, 3); break;
      case 3: r = LDMA0(128, 128, 32, 4); break;
      case 4: r = LDMA(128, 128, 32, 2); break;
      case 8: r = LDMA0(128, 128, 32, 8); break;       // 128 KB ring: sparse launches
#ifdef PNPI_ABLATIONS
      case 11: r = launch_dma<128, 128, 32, 3, 2, 1>(p, grid, st, g_zero_page); break;   // ablation: DMA only
      case 12: r = launch_dma<128, 128, 32, 3, 2, 2>(p, grid, st, g_zero_page); break;   // ablation: compute only
      case 15: r = launch_dma<128, 128, 32, 3, 2, 3>(p, grid, st, g_zero_page); break;   // ablation: activation loads for tap 0 only
#endif
      default: r = LDMA(128, 128, 64, 2); break;
    }
  } else {
    int v64 = g_var64;
    if (v64 == 0 && sparse == 2) v64 = 8;
    else if (v64 == 0 && sparse == 1) v64 = 2;
    switch (v64) {
      case 8: r = LDMA0(64, 64, 64, 8); break;         // 128 KB ring: sparse launches
      case 1: r = LDMA0(64, 64, 64, 2); break;
      case 2: r = LDMA0(64, 64, 64, 4); break;
      case 3: r = LDMA0(64, 64, 32, 4); break;
      default: r = LDMA(64, 64, 64, 3); break;
    }
  }
  if (r) return r;
  if (split > 1) {
    size_t total = (size_t)p.M * ((p.N + 3) / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    const bool vec = p.N % 4 == 0 && p.vt_col0 >= p.N && p.ldo % 4 == 0 && (!p.res || p.ldres % 4 == 0) && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                     ((uintptr_t)p.out & 7) == 0 && (!p.res || ((uintptr_t)p.res & 7) == 0) && ((uintptr_t)ws & 15) == 0;
    // a caller that asked for it gets the slabs instead of the combine launch (alpha == 1: `v * 1 + bias` has the same bits fused or
    // not): it either runs launch_splitk_reduce itself or hands the slabs to a consumer that sums them in the same order (GroupNorm)
    if (vec && deferred && p.alpha == 1.f) { *deferred = p; return (int)hipGetLastError(); }
    if (vec) splitk_reduce_vec_kernel<<<blocks, 256, 0, st>>>(p);
    else splitk_reduce_kernel<<<blocks, 256, 0, st>>>(p);
  }
  return (int)hipGetLastError();
}

// the combine of a launch_igemm call that deferred it (`p` as returned through `deferred`: splitk > 1, the vectorised layout)
int launch_splitk_reduce(const GemmP& p, hipStream_t st) {
  if (p.splitk <= 1 || !p.slab) return -2;
  const size_t total = (size_t)p.M * (p.N / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  splitk_reduce_vec_kernel<<<blocks, 256, 0, st>>>(p);
  return (int)hipGetLastError();
}

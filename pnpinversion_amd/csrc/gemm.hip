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

static constexpr int BK = 64;
static constexpr int LDS_LD = BK + 8;  // halfs

__device__ __forceinline__ void epilogue_store4(const GemmP& p, int m, int nb, const float* v, const float* bias) {
  if (m >= p.M) return;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = nb + j;
    float x = v[j] * p.alpha;
    if (n < p.N) {
      if (bias) x += bias[n];
      if (p.res) x += (float)p.res[(size_t)m * p.ldres + n];
    }
    o[j] = x;
  }
  if (nb + 3 < p.vt_col0 && nb + 3 < p.N && (p.ldo & 3) == 0) {
    half4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = (half_t)o[j];
    *reinterpret_cast<half4*>(p.out + (size_t)m * p.ldo + nb) = h;
    return;
  }
  int b = 0, tok = m;
  if (nb + 3 >= p.vt_col0) { b = m / p.rows_per_batch; tok = m - b * p.rows_per_batch; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = nb + j;
    if (n >= p.N) continue;
    if (n < p.vt_col0) {
      p.out[(size_t)m * p.ldo + n] = (half_t)o[j];
    } else {
      size_t idx = ((size_t)b * (p.N - p.vt_col0) + (n - p.vt_col0)) * p.vt_ld + tok;
      if (p.vt_f32) ((float*)p.outT)[idx] = o[j];
      else ((half_t*)p.outT)[idx] = (half_t)o[j];
    }
  }
}

template <int BM, int BN, bool FASTK>
__global__ void __launch_bounds__(256) igemm_kernel(GemmP p) {
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
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
static int g_tile_order = -1;   // PNPI_TILE_ORDER: force 0 / 1 (ablation)

// ABL: 0 = product kernel; 1 = DMA only (no fragment reads / MFMA); 2 = compute only (no DMA); 3 = activation operand loaded for
// the first filter tap only -- bottleneck ablations for tools/ (1-3 produce garbage).
//
// Tile geometry: WGM x 2 wavefronts (NT = 128 * WGM threads); a wave owns (BM / WGM) x (BN / 2) of the block tile as 32x32 MFMA
// tiles.  Instantiated shapes (BM x BN, wave tile): 64x64 (32x32), 128x128 (64x64), 128x256 (64x128), 128x320 (64x160: 20
// MFMAs per 32-deep k-chunk and barrier, 91 FLOP per operand byte -- the N = 320 / 640 / 1280 layers tile exactly), 256x128.
//
// WK > 1: K-parallel wave groups.  The block holds WK groups of WGM x 2 waves; group g owns its own LDS ring and walks the g-th
// contiguous slice of the block's k-range into its own accumulators (the groups run in lock step on the one workgroup barrier per
// chunk), then groups 1 .. WK-1 hand their fp32 accumulators to group 0 through LDS in a FIXED order and exit; group 0 runs the
// epilogue.  A launch with at most one tile per CU (the one-row inversion forwards, the 16 x 16 / 8 x 8 levels) puts WK times the
// MFMA-issuing waves on every CU this way -- split-K with the partial sums in LDS instead of slabs + a reduce launch.
template <int BM, int BN, int BKT, int NST, int WGM = 2, int ABL = 0, int WK = 1>
struct IgemmGeom {
  static constexpr int NW = 2 * WGM, NT = 64 * NW;              // waves / threads of ONE k-group (= the epilogue's threads)
  static constexpr int NT_ALL = NT * WK;
  static constexpr int WM = BM / WGM, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  static constexpr int RPI = 1024 / (BKT * 2);                 // tile rows per 1-KiB DMA instruction (8 or 16)
  static constexpr int AV = BM / RPI / NW, WV = BN / RPI / NW;  // DMA instructions per wave per chunk and operand
  static constexpr int ROWB = BKT * 2;                          // bytes per tile row
  static constexpr int STAGE = (BM + BN) * ROWB;                // bytes
  static constexpr int RING = NST * STAGE;
  // LDS epilogue: the output tile is staged as [rows][BN + 8] halfs in the idle ring, EPASS passes of EROWS rows (one row of
  // waves per pass when the whole tile does not fit), then written as full rows; G row groups of VPR 16-byte vectors.
  static constexpr int OLD = BN + 8, VPR = BN / 8, G = NT / VPR;
  static constexpr int STATS_B = G * BN * 2 * 4;
  static constexpr int EPASS = (BM * OLD * 2 + STATS_B <= RING) ? 1 : WGM;
  static constexpr int EROWS = BM / EPASS;
  static constexpr int TOLD = EROWS + 8;                        // halfs per staged row of a TRANSPOSED tile ([BN][EROWS + 8])
  static constexpr int EPI_PLAIN = EROWS * OLD * 2 + STATS_B, EPI_TR = BN * TOLD * 2;
  static constexpr int EPI = EPI_PLAIN > EPI_TR ? EPI_PLAIN : EPI_TR;
  static constexpr int RED = (WK - 1) * BM * BN * 4;            // fp32 accumulators of groups 1 .. WK-1 on their way to group 0
  static constexpr int LDS0 = WK * RING > EPI ? WK * RING : EPI;
  static constexpr int LDS = LDS0 > RED ? LDS0 : RED;
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "DMA instructions must divide evenly over the waves");
  static_assert(WM % 32 == 0 && WN % 32 == 0 && BN % 64 == 0, "wave tile");
};

template <int BM, int BN, int BKT, int NST, int WGM = 2, int ABL = 0, int WK = 1>
__global__ void __launch_bounds__(128 * WGM * WK, (IgemmGeom<BM, BN, BKT, NST, WGM, ABL, WK>::LDS * 2 <= 160 * 1024 && BN <= 320 && WK == 1) ? (2 * 2 * WGM / 4) : (2 * WGM * WK / 4))
igemm_dma_kernel(GemmP p, const half_t* __restrict__ zero_page) {
  static_assert(BKT == 64 || BKT == 32, "BKT");
  using GEO = IgemmGeom<BM, BN, BKT, NST, WGM, ABL, WK>;
  constexpr int NT = GEO::NT, NW = GEO::NW;
  constexpr int WM = GEO::WM, WN = GEO::WN, MI = GEO::MI, NI = GEO::NI;
  constexpr int AV = GEO::AV, WV = GEO::WV, ROWB = GEO::ROWB, STAGE = GEO::STAGE;
  constexpr int T32 = 32 * ROWB;                       // bytes per 32-row MFMA tile
  extern __shared__ __attribute__((aligned(1024))) char smem_all[];

  const int tid = threadIdx.x & (NT - 1), lane = tid & 63;                    // thread / wave index inside the k-group
  const int kgrp = WK == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
  char* smem_raw = smem_all + kgrp * GEO::RING;                                 // this group's ring
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // XCD-aware tile order.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with a private L2), so the
  // launch is 1-D and XCD x takes the x-th contiguous eighth of the tile list.  The list is ordered so that the operand
  // that is expensive to re-fetch crosses the fabric once: tile_order 0 = n fastest (an XCD owns a band of m-tiles: the
  // activation is read once, the weight by every XCD), 1 = m fastest (an XCD owns a band of (n, k-split) weight slices:
  // the weight is read once -- the low-resolution 1280-channel convolutions, where the weight is 10x the activation).
  int bx, by, bz;
  {
    const int T = p.gx * p.gy * p.gz, per = (T + 7) >> 3;
    const int tix = p.tile_order == 2 ? (int)blockIdx.x : (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);   // 2: ablation
    if (tix >= T) return;
    if (p.tile_order == 0) { by = tix % p.gy; const int t = tix / p.gy; bx = t % p.gx; bz = t / p.gx; }
    else { bx = tix % p.gx; const int t = tix / p.gx; by = t % p.gy; bz = t / p.gy; }
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int Cin = p.C1 + p.C2;
  const int HoWo = p.Ho * p.Wo;

  // ---- DMA lane roles.  A wave-level DMA instruction fills 1 KiB lane-linearly; the (row, chunk) a lane fetches is the
  // inverse of the swizzled LDS layout:
  //   BKT = 64: 128-byte rows, byte(R, s) = (R>>4)*2048 + (R&7)*256 + ((R>>3)&1)*128 + ((s ^ (R&7)) * 16)
  //   BKT = 32:  64-byte rows, byte(R, s) = (R>>4)*1024 + ((R>>2)&3)*256 + (R&3)*64 + ((s ^ ((R>>2)&3)) * 16)
  // Everything of the gather that does not depend on the k-chunk is precomputed per row; the per-chunk address is pure
  // arithmetic (no selects that could turn into divergent control flow, no runtime-indexed arrays -> no scratch).
  int d_row, d_chunk;   // row within the instruction's row group, logical 16-byte chunk
  if (BKT == 64) {
    const int line_lo = lane >> 4, half = (lane >> 3) & 1;
    d_row = half * 8 + line_lo;         // + 4 * (j & 1) added below
    d_chunk = lane & 7;                 // ^ line below
  } else {
    d_row = (lane >> 4) * 4 + ((lane >> 2) & 3);
    d_chunk = (lane & 3) ^ (lane >> 4);
  }
  int a_y0[AV], a_x0[AV], a_bh[AV], a_chunk[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int j = wave * AV + i;
    int R, ch;
    if (BKT == 64) { const int line = 4 * (j & 1) + (lane >> 4); R = (j >> 1) * 16 + ((lane >> 3) & 1) * 8 + line; ch = (lane & 7) ^ line; }
    else { R = j * 16 + d_row; ch = d_chunk; }
    a_chunk[i] = ch * 8;
    const int m = m0 + R;
    int b = m / HoWo;
    int rr = m - b * HoWo;
    int yo = rr / p.Wo;
    int xo = rr - yo * p.Wo;
    a_bh[i] = b * p.H;
    a_y0[i] = m < p.M ? yo * p.stride - p.pad : -(1 << 20);   // rows past M fail the bounds test for every tap
    a_x0[i] = xo * p.stride - p.pad;
  }
  const int lim_y = p.ups ? 2 * p.H : p.H, lim_x = p.ups ? 2 * p.W : p.W, ups_sh = p.ups ? 1 : 0;
  const int nchunks = p.K / BKT;
  int kc0 = 0, kc1 = nchunks;
  if (p.splitk > 1) {   // kchunks_per_split is given in 64-wide chunks
    kc0 = bz * p.kchunks_per_split * (64 / BKT);
    kc1 = min(nchunks, kc0 + p.kchunks_per_split * (64 / BKT));
  }
  int iters = kc1 > kc0 ? kc1 - kc0 : 0;      // loop trips: the same for every k-group (they share the barrier)
  if (WK > 1) {
    iters = (iters + WK - 1) / WK;
    kc0 = min(kc1, kc0 + kgrp * iters);
    kc1 = min(kc1, kc0 + iters);
  }
  // Weight rows: one pointer per DMA instruction, bumped by BKT per chunk.  Rows past N read the zero page, which is as long
  // as the longest K this kernel is launched with, so they are bumped like the others (no select in the loop).
  const half_t* w_cur[WV];
#pragma unroll
  for (int i = 0; i < WV; ++i) {
    const int j = wave * WV + i;
    int R, ch;
    if (BKT == 64) { const int line = 4 * (j & 1) + (lane >> 4); R = (j >> 1) * 16 + ((lane >> 3) & 1) * 8 + line; ch = (lane & 7) ^ line; }
    else { R = j * 16 + d_row; ch = d_chunk; }
    const int n = n0 + R;
    w_cur[i] = (n < p.N ? p.w + (size_t)n * p.ldw : zero_page) + ch * 8 + kc0 * BKT;
  }

  // Activation rows: the (tap, source, bounds) part of the gather address changes only when the k-chunk walks into a new
  // filter tap or from the first concat source into the second; inside a tap consecutive chunks are consecutive channels.
  // So the full address (bounds test, 64-bit multiply, zero-page select) is rebuilt under a wave-uniform branch on those
  // boundaries only, and the per-chunk work is one pointer bump per DMA instruction.
  int is_tap = (kc0 * BKT) / Cin;              // wave-uniform scalars
  int is_c0 = kc0 * BKT - is_tap * Cin;
  bool retap = true;
  const half_t* a_cur[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) a_cur[i] = zero_page;

  auto issue = [&](int buf) {
    char* sA = smem_raw + buf * STAGE;
    char* sW = sA + BM * ROWB;
    if (retap) {
      int r = 0, s = 0;
      if (p.ksize == 3) { r = is_tap / 3; s = is_tap - 3 * r; }
      const half_t* src; int ld, cc;
      if (is_c0 < p.C1) { src = p.x1; ld = p.ldx1; cc = is_c0; } else { src = p.x2; ld = p.ldx2; cc = is_c0 - p.C1; }
      const long zoff = zero_page - src;   // element distance to the zero page (plain integer arithmetic on addresses)
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const int yi = a_y0[i] + r, xi = a_x0[i] + s;
        const bool ok = (unsigned)yi < (unsigned)lim_y && (unsigned)xi < (unsigned)lim_x;
        const long off = (long)((a_bh[i] + (yi >> ups_sh)) * p.W + (xi >> ups_sh)) * ld + (cc + a_chunk[i]);
        const long mask = -(long)ok;                       // all ones when in range: select without control flow
        a_cur[i] = src + ((off & mask) | ((zoff + a_chunk[i]) & ~mask));
      }
      retap = false;
    }
    if (ABL != 3 || (is_tap == 0)) {   // ABL 3: the activation operand only for the first filter tap (bound of tap-reuse schemes)
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)a_cur[i], (lds_ptr_t)(sA + (wave * AV + i) * 1024), 16, 0, 0);
        a_cur[i] += BKT;
      }
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)w_cur[i], (lds_ptr_t)(sW + (wave * WV + i) * 1024), 16, 0, 0);
      w_cur[i] += BKT;
    }
    is_c0 += BKT;
    if (is_c0 >= Cin) { is_c0 = 0; ++is_tap; retap = true; }
    else if (is_c0 == p.C1) retap = true;
  };

  // Accumulators.  With alpha == 1 the bias is the accumulator's INITIAL value (p.bias_init, set by the launcher): 4 * NI
  // unconditional 16-byte loads per lane, issued before the first DMA instruction and landed long before the first MFMA needs
  // them -- instead of 16 * MI * NI dependent 4-byte loads in the epilogue, each with its own wait, which cost the short-K
  // layers (10 ... 40 k-chunks per tile) more than their whole main loop.  A lane holds columns n .. n + 3 of every 8-column
  // group, the same for every mi.
  floatx16 acc[MI][NI];
  const float* ebias = p.bias_init ? nullptr : p.bias;   // what the epilogue still has to add
  if (p.bias_init && kgrp == 0) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn0 + ni * 32 + 8 * g + 4 * (lane >> 5);
        floatx4 b = *reinterpret_cast<const floatx4*>(p.bias + min(n, p.N - 4));   // N % 4 == 0: a group is all in or all out
        if (n >= p.N) b = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][ni][4 * g + j] = b[j];
      }
#pragma unroll
    for (int mi = 1; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = acc[0][ni];
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  }

  // fragment read addressing (same involution as the DMA source permutation)
  const int lr = lane & 31, hk = lane >> 5;
  int lane_row_off, xk;
  if (BKT == 64) { lane_row_off = (lr >> 4) * 2048 + (lr & 7) * 256 + ((lr >> 3) & 1) * 128; xk = lr & 7; }
  else { lane_row_off = (lr >> 4) * 1024 + ((lr >> 2) & 3) * 256 + (lr & 3) * 64; xk = (lr >> 2) & 3; }

  // NST-deep LDS ring.  Chunk k+NST-1 is issued while chunk k is computed; each wave waits for ITS loads of chunk k with a
  // counted vmcnt (the newer chunks stay in flight across the barrier), then one raw barrier per chunk makes every wave's
  // part of chunk k visible and, at the same time, frees the stage that was read in the previous iteration.
  constexpr int LPS = AV + WV;   // DMA instructions per wave per chunk
  if (iters > 0) {
    const int total = kc1 - kc0;   // this group's chunks (<= iters; a short last group idles through its tail but keeps the barrier)
    int issued = 0;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
      if (issued < total) { if (ABL != 2) issue(st); ++issued; }
    int rd = 0, wr = NST - 1;
    for (int it = 0; it < iters; ++it) {
      const int ahead = issued - it - 1;           // chunks issued after the one needed now (0 .. NST-2)
      static_assert((NST - 2) * LPS <= 63, "vmcnt holds 6 bits");
      if (NST == 2 || ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
      else if (ahead == 3 || NST <= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST > 4 ? 3 : 0) * LPS) : "memory");
      else if (ahead == 4 || NST <= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST > 5 ? 4 : 0) * LPS) : "memory");
      else if (ahead == 5 || NST <= 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST > 6 ? 5 : 0) * LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST > 7 ? 6 : 0) * LPS) : "memory");
      __builtin_amdgcn_s_barrier();
      if (WK > 1 && it >= total) continue;         // wave-uniform: this group has run out of chunks
      if (ABL == 1) {
        if (issued < total) { issue(wr); ++issued; wr = wr + 1 == NST ? 0 : wr + 1; }
        rd = rd + 1 == NST ? 0 : rd + 1;
        continue;
      }
      const char* sA = smem_raw + rd * STAGE + (wm0 >> 5) * T32 + lane_row_off;
      const char* sW = smem_raw + rd * STAGE + BM * ROWB + (wn0 >> 5) * T32 + lane_row_off;
      // Software pipeline inside the chunk: the fragments of k-step kk+1 are read while the MFMAs of k-step kk run, and the
      // first k-step's reads are issued BEFORE the next chunk's DMA instructions (their issue slots -- ~100 cycles each -- then
      // overlap the LDS latency instead of preceding it).  One exposed LDS round trip per chunk instead of one per k-step.
      constexpr int KS = BKT / 16;
      half8 wf[2][NI], af[2][MI];
      auto frag_load = [&](int slot, int kk) {
        const int ko = ((kk * 2 + hk) ^ xk) * 16;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[slot][mi] = *reinterpret_cast<const half8*>(sA + mi * T32 + ko);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[slot][ni] = *reinterpret_cast<const half8*>(sW + ni * T32 + ko);
      };
      frag_load(0, 0);
      if (issued < total) { if (ABL != 2) issue(wr); ++issued; wr = wr + 1 == NST ? 0 : wr + 1; }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        if (kk + 1 < KS) frag_load((kk + 1) & 1, kk + 1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma32(wf[kk & 1][ni], af[kk & 1][mi], acc[mi][ni]);
      }
      rd = rd + 1 == NST ? 0 : rd + 1;
    }
  }

  if (WK > 1) {
    // k-groups 1 .. WK-1 -> group 0, through LDS (the rings are idle: every issued chunk has been waited for and read).  Register r
    // of lane l of wave w travels as one float at [(g-1)][w][tile][r][l]: lane-contiguous, conflict-free; group 0 adds the groups in
    // ascending order, so the sum is bit-reproducible.
    float* sRed = reinterpret_cast<float*>(smem_all);
    constexpr int PER_WAVE = MI * NI * 16 * 64, PER_GRP = NW * PER_WAVE;
    __syncthreads();
    if (kgrp > 0) {
      float* dst = sRed + (size_t)(kgrp - 1) * PER_GRP + wave * PER_WAVE + lane;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((mi * NI + ni) * 16 + r) * 64] = acc[mi][ni][r];
    }
    __syncthreads();
    if (kgrp > 0) return;              // whole waves exit: s_barrier counts the surviving waves only
#pragma unroll
    for (int g = 1; g < WK; ++g) {
      const float* src = sRed + (size_t)(g - 1) * PER_GRP + wave * PER_WAVE + lane;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] += src[((mi * NI + ni) * 16 + r) * 64];
    }
    smem_raw = smem_all;               // the epilogue stages from the start of LDS
  }

  if (p.epi_lds && p.splitk <= 1) {
    // Coalesced epilogue: the output tile is assembled in LDS (the ring is idle now) and written as full rows, 16 bytes per
    // lane.  The residual tile is staged the same way, so out = fp16(alpha*acc + bias + residual) with a single rounding.
    // EPASS passes of EROWS rows (pass e = the waves of wave-row e when the tile is staged in parts).  Store loop: thread t owns
    // the fixed 16-byte vector v = t % VPR of row group g = t / VPR and walks rows g, g + G, ...: per-channel sums for the
    // GroupNorm that consumes this tensor fall out of it (fp32 sums of the rounded fp16 values the consumer will read), and every
    // partial is combined in a FIXED order (rows ascending per thread, then groups ascending): bit-reproducible statistics.
    constexpr int OLD = GEO::OLD, VPR = GEO::VPR, G = GEO::G, EPASS = GEO::EPASS, EROWS = GEO::EROWS;
    half_t* sOut = reinterpret_cast<half_t*>(smem_raw);
    const bool tr_tile = p.outT != nullptr && n0 >= p.vt_col0;      // block-uniform (the launcher aligns vt_col0 to the tile width)
    const int sv = tid % VPR, sg = tid / VPR;
    const bool s_active = sg < G;
    float cs[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] = 0.f; cq[j] = 0.f; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPASS; ++e) {
      const int r0 = e * EROWS;                         // first tile row of this pass
      const bool res_staged = p.res && !p.res_late;
      if (res_staged) {
        if (s_active) {
          // clamped addresses instead of predicated loads: the loads of an unrolled group issue back to back (rows past M / columns
          // past N stage finite junk that is never stored)
          const int nres = min(n0 + sv * 8, p.N - 8);
#pragma unroll 8
          for (int r = sg; r < EROWS; r += G) {
            const int m = min(m0 + r0 + r, p.M - 1);
            *reinterpret_cast<half8*>(sOut + r * OLD + sv * 8) = ldg_half8(p.res + (size_t)m * p.ldres + nres);
          }
        }
        __syncthreads();
      }
      if (tr_tile) {
        // V^T-style output (columns >= vt_col0 are written transposed per batch item): the tile is staged TRANSPOSED, [BN][EROWS + 8],
        // and leaves as 16-byte runs of 8 consecutive tokens of one output channel
        constexpr int TOLD = GEO::TOLD, VPT = EROWS / 8;
        if (EPASS == 1 || (wave >> 1) == e) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int ml = wm0 - r0 + mi * 32 + (lane & 31);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int nl = wn0 + ni * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  float o = acc[mi][ni][4 * g + j] * p.alpha;
                  if (ebias && n0 + nl + j < p.N) o += ebias[n0 + nl + j];
                  sOut[(nl + j) * TOLD + ml] = (half_t)o;
                }
              }
            }
          }
        }
        __syncthreads();
        const int ncol = p.N - p.vt_col0;
        for (int idx = tid; idx < BN * VPT; idx += NT) {
          const int nl = idx / VPT, v = idx - nl * VPT;
          const int n = n0 + nl, m = m0 + r0 + v * 8;
          if (n < p.N && m < p.M) {
            const int b = m / p.rows_per_batch, tok = m - b * p.rows_per_batch;
            half_t* dst = (half_t*)p.outT + ((size_t)b * ncol + (n - p.vt_col0)) * p.vt_ld + tok;
            *reinterpret_cast<half8*>(dst) = *reinterpret_cast<const half8*>(sOut + nl * TOLD + v * 8);
          }
        }
        if (e + 1 < EPASS) __syncthreads();
        continue;
      }
      if (EPASS == 1 || (wave >> 1) == e) {
        // two copies, selected by a block-uniform branch: with the bias already in the accumulators (the usual case) the staging
        // loop has no per-element control flow at all
        auto stage = [&](auto has_bias) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int ml = wm0 - r0 + mi * 32 + (lane & 31);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              half4* slot0 = reinterpret_cast<half4*>(sOut + ml * OLD + wn0 + ni * 32 + 4 * (lane >> 5));   // slot of group g: + 2 * g
              half4 r4[4];
              if (res_staged) {                  // the four residual slots of this 32-column tile in flight together
#pragma unroll
                for (int g = 0; g < 4; ++g) r4[g] = slot0[2 * g];
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + ni * 32 + 8 * g + 4 * (lane >> 5);
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  o[j] = acc[mi][ni][4 * g + j] * p.alpha;
                  if constexpr (decltype(has_bias)::value) { if (n + j < p.N) o[j] += ebias[n + j]; }
                }
                if (res_staged) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) o[j] += (float)r4[g][j];
                }
                half4 h4 = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                slot0[2 * g] = h4;
              }
            }
          }
        };
        if (ebias) stage(std::true_type{}); else stage(std::false_type{});
      }
      __syncthreads();
      if (!p.geglu) {
        const int n = n0 + sv * 8;
        if (s_active && n < p.N) {
#pragma unroll 4
          for (int r = sg; r < EROWS; r += G) {
            const int m = m0 + r0 + r;
            half8 val = *reinterpret_cast<const half8*>(sOut + r * OLD + sv * 8);
            if (m < p.M) {
              if (p.res && p.res_late) {   // residual added on the way out (coalesced 16-byte reads, no staging pass / barrier)
                const half8 rv = ldg_half8(p.res + (size_t)m * p.ldres + n);
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = (half_t)((float)val[j] + (float)rv[j]);
              }
              *reinterpret_cast<half8*>(p.out + (size_t)m * p.ldo + n) = val;
              if (p.stats) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = (float)val[j]; cs[j] += f; cq[j] += f * f; }
              }
            }
          }
        }
      } else {
        // columns come in [x(32) | gate(32)] groups; the block's BN columns hold BN/2 outputs starting at column n0/2
        constexpr int VPO = BN / 16;
        for (int idx = tid; idx < EROWS * VPO; idx += NT) {
          const int r = idx / VPO, v = idx - r * VPO;
          const int m = m0 + r0 + r;
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
      if (e + 1 < EPASS) __syncthreads();               // the next pass overwrites the staging rows
    }
    if (p.stats && !p.geglu) {
      float* sSt = reinterpret_cast<float*>(sOut + EROWS * OLD);       // [G][BN][2]
      if (s_active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sSt[(sg * BN + sv * 8 + j) * 2 + 0] = cs[j];
          sSt[(sg * BN + sv * 8 + j) * 2 + 1] = cq[j];
        }
      }
      __syncthreads();
      for (int c = tid; c < BN; c += NT) {
        if (n0 + c < p.N) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) { s += sSt[(g * BN + c) * 2]; q += sSt[(g * BN + c) * 2 + 1]; }
          float* dst = p.stats + ((size_t)bx * p.N + n0 + c) * 2;
          dst[0] = s;
          dst[1] = q;
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
            float* dst = p.slab + ((size_t)bz * p.M + m) * p.N + nb;
            if (nb + 3 < p.N) *reinterpret_cast<floatx4*>(dst) = floatx4{v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (nb + j < p.N) dst[j] = v[j];
            }
          }
        } else {
          epilogue_store4(p, m, nb, v, ebias);
        }
      }
    }
  }
}

static int g_last_geom[7] = {0, 0, 0, 0, 0, 0, 0};   // template arguments of the most recent igemm_dma_kernel launch (all 0: the v1 kernel)
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
  static unsigned long long attr_devs = 0;
  if (first_on_device(attr_devs)) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_dma_kernel<BM, BN, BKT, NST, WGM, ABL, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  igemm_dma_kernel<BM, BN, BKT, NST, WGM, ABL, WK><<<grid, GEO::NT_ALL, lds, st>>>(p, zero_page);
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
  p.out = nullptr; p.ldo = 0; p.outT = nullptr; p.vt_col0 = 1 << 30; p.vt_ld = 0; p.vt_f32 = 0; p.rows_per_batch = 1;
  p.slab = nullptr; p.splitk = 1; p.kchunks_per_split = 0; p.geglu = 0; p.epi_lds = 0; p.stats = nullptr; p.res_late = 0; p.bias_init = 0;
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
static int g_force_split = 0;   // > 0 with igemm_force_cfg: split-K of every auto-configured launch (in-forward tuning sweeps)
static int g_last_cfg = -1, g_last_split = 1;   // what the most recent launch_igemm used (profiling dumps)
void igemm_last_launch(int* cfg, int* split, int* geom) { *cfg = g_last_cfg; *split = g_last_split; for (int i = 0; i < 7; ++i) geom[i] = g_last_geom[i]; }
static int g_force_cfg = -1;   // >= 0: every auto-configured launch uses this tile configuration (tests, whole-forward A/B)
static int g_var128 = 2, g_var64 = 0, g_var256 = 0, g_var320 = 1, g_var256n = 1;
static long g_v128_bk64_tiles = 0;   // PNPI_V128_BK64_TILES: tile count from which the 128x128 kernel switches to 128-byte rows   // tuning variants (PNPI_IGEMM_V128 / PNPI_IGEMM_V64)
void igemm_set_dma(int on) { g_use_dma = on; }
// process-wide tuning knobs (A/B measurements inside one process, tests of the non-default variants); 0 on success
int igemm_set_tuning(const char* key, int v) {
  struct { const char* k; int* p; } tab[] = {{"igemm_dma", &g_use_dma}, {"igemm_v128", &g_var128}, {"igemm_v64", &g_var64}, {"igemm_v256", &g_var256},
                                             {"igemm_v320", &g_var320}, {"igemm_v256n", &g_var256n}, {"igemm_wide", &g_wide}, {"tile_order", &g_tile_order}, {"igemm_force_cfg", &g_force_cfg}, {"igemm_force_split", &g_force_split}, {"igemm_bias_init", &g_bias_init}, {"igemm_deep_rings", &g_deep_rings}, {"igemm_vt_lds", &g_vt_lds}, {"igemm_res_late", &g_res_late}, {"igemm_table", &g_use_table}, {"igemm_table_near", &g_table_near}};
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
// 13 = 128x128 with a two-stage ring; 14 / 15 = 128x128 / 128x256 with 128-byte rows: reached through the table or force_cfg

int launch_igemm(GemmP p, float* ws, size_t ws_bytes, hipStream_t st, int force_cfg, int force_split, int* cfg_used,
                 int* stats_tile_rows) {
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
        const int ebn = (e.cfg == 4 || e.cfg == 6 || e.cfg == 12) ? 320 : ((e.cfg == 5 || e.cfg == 7 || e.cfg == 15) ? 256 : ((e.cfg == 1 || e.cfg == 8 || e.cfg == 11) ? 64 : 128));
        const bool split_ok = e.split == 1 || (!p.geglu && ws && (size_t)e.split * p.M * p.N * sizeof(float) <= ws_bytes);
        const bool vt_ok = p.vt_col0 >= p.N || p.vt_col0 % ebn == 0;
        if (split_ok && vt_ok && (g_wide || e.cfg < 4 || e.cfg == 13 || e.cfg == 14)) { cfg = e.cfg; split = e.split; }
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
  if (cfg >= 8 && !dma_ok) cfg = (cfg == 8 || cfg == 11) ? 1 : 0;
  if (cfg >= 3 && !dma_ok) cfg = 0;                                 // 256x128 / 128x320 / 128x256 exist only as LDS-DMA kernels
  // whatever chose the split (cost model or a caller-forced value): the slabs must fit the workspace and each split needs work
  if (split > 1) {
    if (split > nchunks) split = nchunks;
    if (ws == nullptr || (size_t)split * p.M * p.N * sizeof(float) > ws_bytes || p.geglu || cfg == 3) split = 1;
  }
  const bool c64 = cfg == 1 || cfg == 8 || cfg == 11 || cfg == 12, c256m = cfg == 3 || cfg == 6 || cfg == 7;
  if (cfg_used) *cfg_used = split > 1 ? 2 : (((cfg >= 4 && cfg <= 7) || cfg == 12 || cfg == 15) ? 9 : (c64 ? 1 : 0));
  p.splitk = split;
  p.kchunks_per_split = (nchunks + split - 1) / split;
  g_last_cfg = cfg; g_last_split = split;
  for (int i = 0; i < 7; ++i) g_last_geom[i] = 0;
  const int bn_sel = (cfg == 4 || cfg == 6 || cfg == 12) ? 320 : ((cfg == 5 || cfg == 7 || cfg == 15) ? 256 : (c64 ? 64 : 128));
  const bool vt_none = p.vt_col0 >= p.N;
  // transposed (V^T) columns through the LDS epilogue too, when whole tiles are either plain or transposed and 8-token runs stay
  // inside one batch item
  const bool vt_lds = !vt_none && dma_ok && p.outT && !p.vt_f32 && p.vt_col0 % bn_sel == 0 && p.rows_per_batch % 8 == 0 && p.vt_ld % 8 == 0 &&
                      p.M % 8 == 0 && !p.res && !p.geglu && g_vt_lds;
  p.epi_lds = (vt_none || vt_lds) && (p.N % 8 == 0) && (p.vt_col0 == 0 || p.ldo % 8 == 0) && (!p.res || p.ldres % 8 == 0) && split == 1;
  if (vt_lds) p.stats = nullptr;
  p.res_late = g_res_late;
  p.bias_init = (dma_ok && split == 1 && p.bias && p.alpha == 1.f && p.N % 4 == 0 && ((uintptr_t)p.bias & 15) == 0 && g_bias_init) ? 1 : 0;
  if (p.geglu && !(p.epi_lds && dma_ok && p.N % 64 == 0)) return -7;   // GEGLU exists only in the LDS epilogue
  if (!(p.epi_lds && dma_ok && !p.geglu)) p.stats = nullptr;       // statistics come only from the DMA kernel's LDS epilogue
  const int bm = c256m ? 256 : (c64 ? 64 : 128);
  const int bn = bn_sel;
  if (stats_tile_rows) *stats_tile_rows = p.stats ? bm : 0;
  p.slab = ws;
  dim3 grid((p.M + bm - 1) / bm, (p.N + bn - 1) / bn, split);
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
    if (sparse == 2 && g_var320 == 1) r = launch_dma<128, 320, 32, 5>(p, grid, st, g_zero_page);
    else switch (g_var320) {
      case 0: r = launch_dma<128, 320, 32, 3>(p, grid, st, g_zero_page); break;       // 84 KB ring = whole-tile epilogue, 1 block / CU
      case 2: r = launch_dma<128, 320, 64, 2>(p, grid, st, g_zero_page); break;       // 128-byte rows, 112 KB, 1 block / CU
      case 11: r = launch_dma<128, 320, 32, 2, 2, 1>(p, grid, st, g_zero_page); break;   // ablation: DMA only
      case 12: r = launch_dma<128, 320, 32, 2, 2, 2>(p, grid, st, g_zero_page); break;   // ablation: compute only
      default: r = launch_dma<128, 320, 32, 2>(p, grid, st, g_zero_page); break;      // 56 KB ring, two-pass epilogue, 2 blocks / CU
    }
  } else if (cfg == 12) {
    r = launch_dma<64, 320, 32, 2>(p, grid, st, g_zero_page);             // 64 x 320: 768 tiles on the 12-row 64 x 64 level = 3 per CU
  } else if (cfg == 13) {
    r = launch_dma<128, 128, 32, 2>(p, grid, st, g_zero_page);            // 32 KB ring (two-pass epilogue): 4 blocks / CU for the short-K layers
  } else if (cfg == 14) {
    r = launch_dma<128, 128, 64, 2>(p, grid, st, g_zero_page);            // 128-byte rows, half the barriers per k: 64 KB, 2 blocks / CU
  } else if (cfg == 15) {
    r = launch_dma<128, 256, 64, 2>(p, grid, st, g_zero_page);            // the same for the 128 x 256 tile: 96 KB, 1 block / CU
  } else if (cfg == 8) {
    r = launch_dma<64, 64, 64, 2, 2, 0, 4>(p, grid, st, g_zero_page);      // 16 waves: 4 k-groups
  } else if (cfg == 11) {
    r = launch_dma<64, 64, 64, 2, 2, 0, 2>(p, grid, st, g_zero_page);      // 8 waves: 2 k-groups, 64 KB (2 blocks / CU)
  } else if (cfg == 9) {
    r = launch_dma<128, 128, 32, 3, 2, 0, 2>(p, grid, st, g_zero_page);    // 8 waves: 2 k-groups, 96 KB
  } else if (cfg == 10) {
    r = launch_dma<128, 128, 64, 2, 2, 0, 2>(p, grid, st, g_zero_page);    // 8 waves: 2 k-groups, 128-byte rows, 128 KB
  } else if (cfg == 6) {
    r = launch_dma<256, 320, 64, 2, 4>(p, grid, st, g_zero_page);
  } else if (cfg == 7) {
    r = launch_dma<256, 256, 64, 2, 4>(p, grid, st, g_zero_page);
  } else if (cfg == 5) {
    if (sparse == 2 && g_var256n == 1) r = launch_dma<128, 256, 32, 6>(p, grid, st, g_zero_page);
    else switch (g_var256n) {
      case 0: r = launch_dma<128, 256, 32, 3>(p, grid, st, g_zero_page); break;       // 72 KB, 2 blocks / CU
      case 2: r = launch_dma<128, 256, 64, 2>(p, grid, st, g_zero_page); break;
      default: r = launch_dma<128, 256, 32, 2>(p, grid, st, g_zero_page); break;      // 48 KB: two-pass epilogue, 3 blocks / CU by LDS
    }
  } else if (cfg == 0) {
    int var = g_var128;
    // 128-byte rows with 2 stages (64 KB, 2 blocks / CU) beat 64-byte rows with 3 stages (48 KB, 3 blocks / CU) once every CU
    // holds two blocks that cover for each other's exposed loads; below that the deeper ring wins
    if (g_v128_bk64_tiles > 0 && var == 2 && (long)grid.x * grid.y * grid.z >= g_v128_bk64_tiles) var = 0;
    if (var == 2 && sparse == 2) var = 8;
    else if (var == 2 && sparse == 1) var = 3;
    switch (var) {
      case 1: r = launch_dma<128, 128, 64, 3>(p, grid, st, g_zero_page); break;
      case 2: r = launch_dma<128, 128, 32, 3>(p, grid, st, g_zero_page); break;
      case 3: r = launch_dma<128, 128, 32, 4>(p, grid, st, g_zero_page); break;
      case 4: r = launch_dma<128, 128, 32, 2>(p, grid, st, g_zero_page); break;
      case 8: r = launch_dma<128, 128, 32, 8>(p, grid, st, g_zero_page); break;       // 128 KB ring: sparse launches
      case 11: r = launch_dma<128, 128, 32, 3, 2, 1>(p, grid, st, g_zero_page); break;   // ablation: DMA only
      case 12: r = launch_dma<128, 128, 32, 3, 2, 2>(p, grid, st, g_zero_page); break;   // ablation: compute only
      case 15: r = launch_dma<128, 128, 32, 3, 2, 3>(p, grid, st, g_zero_page); break;   // ablation: activation loads for tap 0 only
      default: r = launch_dma<128, 128, 64, 2>(p, grid, st, g_zero_page); break;
    }
  } else {
    int v64 = g_var64;
    if (v64 == 0 && sparse == 2) v64 = 8;
    else if (v64 == 0 && sparse == 1) v64 = 2;
    switch (v64) {
      case 8: r = launch_dma<64, 64, 64, 8>(p, grid, st, g_zero_page); break;         // 128 KB ring: sparse launches
      case 1: r = launch_dma<64, 64, 64, 2>(p, grid, st, g_zero_page); break;
      case 2: r = launch_dma<64, 64, 64, 4>(p, grid, st, g_zero_page); break;
      case 3: r = launch_dma<64, 64, 32, 4>(p, grid, st, g_zero_page); break;
      default: r = launch_dma<64, 64, 64, 3>(p, grid, st, g_zero_page); break;
    }
  }
  if (r) return r;
  if (split > 1) {
    size_t total = (size_t)p.M * ((p.N + 3) / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    const bool vec = p.N % 4 == 0 && p.vt_col0 >= p.N && p.ldo % 4 == 0 && (!p.res || p.ldres % 4 == 0) && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                     ((uintptr_t)p.out & 7) == 0 && (!p.res || ((uintptr_t)p.res & 7) == 0) && ((uintptr_t)ws & 15) == 0;
    if (vec) splitk_reduce_vec_kernel<<<blocks, 256, 0, st>>>(p);
    else splitk_reduce_kernel<<<blocks, 256, 0, st>>>(p);
  }
  return (int)hipGetLastError();
}

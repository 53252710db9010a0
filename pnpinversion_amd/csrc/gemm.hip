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
#include "ops.h"

static constexpr int BK = 64;
static constexpr int LDS_LD = BK + 8;  // halfs

__device__ __forceinline__ void epilogue_store4(const GemmP& p, int m, int nb, const float* v) {
  if (m >= p.M) return;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = nb + j;
    float x = v[j] * p.alpha;
    if (n < p.N) {
      if (p.bias) x += p.bias[n];
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
          epilogue_store4(p, m, nb, v);
        }
      }
    }
  }
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
    epilogue_store4(p, m, nb, v);
  }
}

void gemm_defaults(GemmP& p) {
  p.x1 = nullptr; p.x2 = nullptr; p.C1 = 0; p.C2 = 0; p.ldx1 = 0; p.ldx2 = 0;
  p.B = 1; p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.ksize = 1; p.stride = 1; p.pad = 0; p.ups = 0;
  p.w = nullptr; p.ldw = 0; p.M = 0; p.N = 0; p.K = 0; p.bias = nullptr; p.res = nullptr; p.ldres = 0; p.alpha = 1.f;
  p.out = nullptr; p.ldo = 0; p.outT = nullptr; p.vt_col0 = 1 << 30; p.vt_ld = 0; p.vt_f32 = 0; p.rows_per_batch = 1;
  p.slab = nullptr; p.splitk = 1; p.kchunks_per_split = 0;
}

static constexpr size_t lds_bytes(int BM, int BN) { return (size_t)(2 * BM + 2 * BN) * LDS_LD * sizeof(half_t); }

int igemm_init() {
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(128, 128)));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(128, 128)));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<64, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(64, 64)));
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<64, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(64, 64)));
  return 0;
}

int launch_igemm(GemmP p, float* ws, size_t ws_bytes, hipStream_t st, int force_cfg, int force_split, int* cfg_used) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -2;
  const int Cin = p.C1 + p.C2;
  if ((Cin & 7) || (p.K & 7) || (p.C1 & 7) || (p.ldw & 7) || (p.ldx1 & 7) || (p.C2 && (p.ldx2 & 7))) return -3;
  if (p.K != p.ksize * p.ksize * Cin) return -4;
  if (p.vt_col0 > p.N) p.vt_col0 = p.N;
  const bool fast = (Cin % BK == 0) && (p.C1 % BK == 0);
  const int nchunks = (p.K + BK - 1) / BK;
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  int cfg = force_cfg;
  int split = 1;
  if (cfg < 0) {
    if (t128 >= 192) cfg = 0;
    else if (t64 >= 160) cfg = 1;
    else cfg = 2;
  }
  if (cfg == 2) {
    split = force_split > 0 ? force_split : (int)((512 + t64 - 1) / t64);
    if (split > 16) split = 16;
    if (split > nchunks) split = nchunks;
    // every slice must own at least 4 k-chunks, otherwise the reduction traffic dominates
    while (split > 1 && nchunks / split < 4) --split;
    size_t need = (size_t)split * p.M * p.N * sizeof(float);
    if (split > 1 && (ws == nullptr || need > ws_bytes)) split = 1;
  }
  if (cfg_used) *cfg_used = cfg == 0 ? 0 : (split > 1 ? 2 : 1);
  p.splitk = split;
  p.kchunks_per_split = (nchunks + split - 1) / split;
  p.slab = ws;
  if (cfg == 0) {
    dim3 grid((p.M + 127) / 128, (p.N + 127) / 128, 1);
    if (fast) igemm_kernel<128, 128, true><<<grid, 256, lds_bytes(128, 128), st>>>(p);
    else igemm_kernel<128, 128, false><<<grid, 256, lds_bytes(128, 128), st>>>(p);
  } else {
    dim3 grid((p.M + 63) / 64, (p.N + 63) / 64, split);
    if (fast) igemm_kernel<64, 64, true><<<grid, 256, lds_bytes(64, 64), st>>>(p);
    else igemm_kernel<64, 64, false><<<grid, 256, lds_bytes(64, 64), st>>>(p);
    if (split > 1) {
      size_t total = (size_t)p.M * ((p.N + 3) / 4);
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      splitk_reduce_kernel<<<blocks, 256, 0, st>>>(p);
    }
  }
  return (int)hipGetLastError();
}

// Microbenchmark: do the matrix pipe and the VALU of one SIMD overlap when two DIFFERENT waves feed them (the flash-attention question: a
// wave's softmax -- v_exp_f32 / v_max / v_cvt_pk on the score accumulators -- under another wave's MFMAs)?  One workgroup of 8 waves per CU:
// waves 0-3 (one per SIMD) issue v_mfma_f32_32x32x16_f16 back to back, waves 4-7 (the other wave of each SIMD) a softmax-like VALU stream on
// registers (per "tile": 32 v_exp_f32, 32 v_max_f32, 16 v_cvt_pk, 32 v_fma -- the per-lane work of a 64-key tile).  MODE 1 = MFMA waves alone,
// 2 = VALU waves alone, 3 = both.  both ~ max(alone) -> the pipes overlap across waves; both ~ sum -> the SIMD serialises them.
// VGPRFORM=1 (build with -mllvm -amdgpu-mfma-vgpr-form=1): the accumulators live in VGPRs as in attn.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE, int SAMEWAVE>
__global__ void __launch_bounds__(512, 2) k(int tiles, float* sink, float seed) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mfma_wave = wave < 4;
  if (SAMEWAVE) {          // one wave per SIMD doing both, alternating like the kernel: 14 MFMAs then the softmax block
    if (!mfma_wave) return;
    floatx16 acc[2] = {}, s[2] = {};
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane & 7)); b[j] = (_Float16)(0.02f * j); }
    float m = seed;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int i = 0; i < 6; ++i) s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, s[i & 1], 0, 0, 0);
      half2_t pk[16];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float e0 = __builtin_amdgcn_exp2f(s[r][i] - m), e1 = __builtin_amdgcn_exp2f(s[r][i + 1] - m);
          m = fmaxf(m, fmaxf(s[r][i], s[r][i + 1]) * 1e-6f);
          pk[r * 8 + i / 2] = half2_t{(_Float16)e0, (_Float16)e1};
        }
      half8 p0, p1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { p0[2 * j] = pk[j][0]; p0[2 * j + 1] = pk[j][1]; p1[2 * j] = pk[4 + j][0]; p1[2 * j + 1] = pk[4 + j][1]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, (i & 2) ? p0 : p1, acc[i & 1], 0, 0, 0);
    }
    float r = m;
    for (int i = 0; i < 16; ++i) r += acc[0][i] + acc[1][i];
    if (r == 12345.f) sink[blockIdx.x] = r;
    return;
  }
  if (mfma_wave) {
    if (!(MODE & 1)) return;
    floatx16 acc[2] = {};
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane & 7)); b[j] = (_Float16)(0.02f * j); }
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int i = 0; i < 14; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 1], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[0][i] + acc[1][i];
    if (r == 12345.f) sink[blockIdx.x] = r;
  } else {
    if (!(MODE & 2)) return;
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = seed * (float)(i + lane);
    float m = seed;
    for (int t = 0; t < tiles; ++t) {
      half2_t pk[16];
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        const float e0 = __builtin_amdgcn_exp2f(x[i] - m), e1 = __builtin_amdgcn_exp2f(x[i + 1] - m);
        m = fmaxf(m, fmaxf(x[i], x[i + 1]) * 1e-6f);
        pk[i / 2] = half2_t{(_Float16)e0, (_Float16)e1};
        x[i] = x[i] * 0.999f + (float)pk[i / 2][0];
        x[i + 1] = x[i + 1] * 0.999f + (float)pk[i / 2][1];
      }
    }
    float r = m;
    for (int i = 0; i < 32; ++i) r += x[i];
    if (r == 12345.f) sink[blockIdx.x] = r;
  }
}
// One wave per SIMD issuing BOTH streams, independent of each other, interleaved in program order: FILL VALU instructions (exp2 + convert
// on registers the MFMAs do not touch) after every MFMA.  FILL = 0: the MFMAs alone; -1: the VALU alone.
template <int FILL>
__global__ void __launch_bounds__(256) k_inwave(int tiles, float* sink, float seed) {
  const int lane = threadIdx.x & 63;
  floatx16 acc[2] = {};
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane & 7)); b[j] = (_Float16)(0.02f * j); }
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = seed * (float)(i + lane);
  for (int t = 0; t < tiles; ++t) {
    if (FILL >= 0) {
#pragma unroll
      for (int i = 0; i < 14; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 1], 0, 0, 0);
    }
    if (FILL != 0) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        const float e0 = __builtin_amdgcn_exp2f(x[i]), e1 = __builtin_amdgcn_exp2f(x[i + 1]);
        const half2_t pk = half2_t{(_Float16)e0, (_Float16)e1};
        x[i] = (float)pk[0] * 0.5f; x[i + 1] = (float)pk[1] * 0.5f;
      }
    }
    if (FILL > 0) {
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
      }
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[0][i] + acc[1][i];
  for (int i = 0; i < 32; ++i) r += x[i];
  if (r == 12345.f) sink[blockIdx.x] = r;
}
template <int FILL>
static float run_inwave(int tiles, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_inwave<FILL><<<256, 256>>>(64, sink, 0.001f); hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0); k_inwave<FILL><<<256, 256>>>(tiles, sink, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best * 1e3f;
}

template <int MODE, int SW>
static float run(int tiles, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, SW><<<256, 512>>>(64, sink, 0.001f); hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0); k<MODE, SW><<<256, 512>>>(tiles, sink, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best * 1e3f;
}
int main() {
  float* sink; hipMalloc(&sink, 4096);
  const int tiles = 4096;
  const float m = run<1, 0>(tiles, sink), v = run<2, 0>(tiles, sink), b = run<3, 0>(tiles, sink), sw = run<3, 1>(tiles, sink);
  printf("%d tiles per wave (14 MFMA 32x32x16 = 448 matrix-pipe cycles; 32 exp2 + 32 max + 16 cvt_pk + 64 fma per lane)\n", tiles);
  printf("MFMA waves alone %.1f us (%.0f cycles / tile at 2.4 GHz), VALU waves alone %.1f us (%.0f), both %.1f us: both / max = %.2f, both / sum = %.2f\n", m,
         m * 2400 / tiles, v, v * 2400 / tiles, b, b / (m > v ? m : v), b / (m + v));
  printf("one wave per SIMD doing both in turn (6 MFMA, softmax block, 8 MFMA): %.1f us (%.0f cycles / tile)\n", sw, sw * 2400 / tiles);
  {
    const float m0 = run_inwave<0>(tiles, sink), v0 = run_inwave<-1>(tiles, sink), b6 = run_inwave<6>(tiles, sink), b3 = run_inwave<3>(tiles, sink);
    printf("ONE wave per SIMD, independent streams in program order (14 MFMA + 32 exp2 + 16 cvt_pk + 32 mul per tile): MFMA alone %.0f cycles / tile, VALU alone %.0f, "
           "interleaved 6 VALU per MFMA %.0f, 3 per MFMA %.0f\n", m0 * 2400 / tiles, v0 * 2400 / tiles, b6 * 2400 / tiles, b3 * 2400 / tiles);
  }
  printf("%s\n", hipGetLastError() == hipSuccess ? "ok" : "ERR");
  return 0;
}

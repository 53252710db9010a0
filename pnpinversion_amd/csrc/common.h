// Shared device/host definitions for the pnpi gfx950 kernels.
// CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x16_f16, 160 KiB LDS per CU.
#pragma once
#include <atomic>
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define PNPI_WAVE 64

// D = A(32 x 16) * B(16 x 32) + C, f16 inputs, f32 accumulate.
//   A operand: lane l holds row (l & 31), k-slots (l >> 5) * 8 + [0, 8)
//   B operand: lane l holds col (l & 31), k-slots (l >> 5) * 8 + [0, 8)
//   C/D      : lane l holds col (l & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r in [0, 16)
// The hardware contracts over matching (lane >> 5, slot) pairs, so any k permutation that is applied
// identically to both operands is legal; the attention kernels rely on that.
__device__ __forceinline__ floatx16 mfma32(half8 a, half8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ half8 zero_half8() {
  half8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (half_t)0.f;
  return z;
}

__device__ __forceinline__ half8 ldg_half8(const half_t* p) { return *reinterpret_cast<const half8*>(p); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// Exact-form (erf) GELU of torch.nn.functional.gelu, x * Phi(x), branch-free: Phi(-|x|) = erfc(|x| / sqrt 2) / 2 by Abramowitz &
// Stegun 7.1.26 (|error| <= 1.5e-7 absolute -- 4000x below an fp16 ulp of the product) on the raw v_rcp_f32 / v_exp_f32; the
// library erff costs several hundred cycles per wavefront with its range branches, and the fused GEGLU epilogue evaluates one per
// output element (63 M per 64 x 64-level feed-forward at 12 rows).
__device__ __forceinline__ float gelu_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float half_erfc = 0.5f * p * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);   // Phi(-|x|)
  return x * (x >= 0.f ? 1.0f - half_erfc : half_erfc);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// hipFuncSetAttribute (dynamic-LDS opt-in) is per DEVICE: a process that drives a second device must set it there too, and two
// contexts may launch the same template instance from two threads (stage-overlapped sweeps: ctypes releases the GIL).  `fn` runs once
// per device under a mutex; the device's bit is published only after it succeeded, so no thread can launch before the attribute is set.
struct DeviceOnce {
  std::mutex mu;
  std::atomic<unsigned long long> done{0};
};
template <class F>
static inline int once_per_device(DeviceOnce& o, F&& fn) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return fn();
  if (o.done.load(std::memory_order_acquire) >> dev & 1ull) return 0;
  std::lock_guard<std::mutex> g(o.mu);
  if (o.done.load(std::memory_order_relaxed) >> dev & 1ull) return 0;
  const int r = fn();
  if (r == 0) o.done.fetch_or(1ull << dev, std::memory_order_release);
  return r;
}

#define HIP_CHECK_RET(expr)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return (int)_e;           \
  } while (0)

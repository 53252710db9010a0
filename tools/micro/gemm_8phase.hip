// The guide's "256^2 8-phase template" (cdna_hip_programming.md, "The 256^2 8-phase template"), written from its description for
// fp16 as a plain NT GEMM  C[M][N] = A[M][K] . B[N][K]^T  -- VERDICT r5 item 3: run it in this tree, on these boxes, on random operands,
// interleaved with igemm_pp<256,256> (tools/gemm_8phase_ab.py drives both in one process).
//
//   tile 256 x 256, K-step 64, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 fragments of v_mfma_f32_16x16x32_f16;
//   LDS 128 KiB = 2 K-tile buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 64 halfs (16 KiB): A half h = rows wr * 128 + h * 64 + [0, 64)
//   of both wave rows, B half h = columns wc * 64 + h * 32 + [0, 32) of the four wave columns -- what ONE phase of every wave reads;
//   layout st_16x32: 1-KiB subtiles of 16 rows x 32 halfs (64-byte rows), byte ^= ((byte >> 9) & 1) << 5 inside a subtile; one
//   LDS-DMA wave instruction (64 lanes x 16 B, lane-linear destination) fills one subtile, the permutation is applied to its source;
//   4 phases per K-tile, one C quadrant (64 x 32 x 64 = 16 MFMAs) each:  p1 A0 x B0 (reads A0, B0)   p2 A0 x B1 (reads B1)
//   p3 A1 x B1 (reads A1)   p4 A1 x B0 (reads B0 again);  every phase stages ONE half-tile (2 DMA instructions per wave), one phase
//   after the last read of the region it overwrites:  p1 B0(t+1)  p2 A0(t+2)  p3 B1(t+2)  p4 A1(t+2);
//   s_waitcnt vmcnt(6) ONCE per K-tile (phase 4: three half-tiles stay in flight), never 0 in the loop; the wave row wr = 1 runs one
//   barrier behind wr = 0 (its read / stage segment beside the other row's MFMA segment on the same SIMD).
// VAR bit 0: the phase's s_waitcnt lgkmcnt(0) BEFORE its first barrier instead of after it (the conservative WAR form).
// VAR bit 1 (ABLATION, wrong results): the A operand is staged for K-tiles 0 and 1 only -- the upper bound of what an LDS halo tile for the
//            3 x 3 convolutions can return (activation DMA once per channel slab instead of once per tap; here: never).
// VAR bit 2 (ABLATION, wrong results): the same for the B operand.
//
// Build (shared library for the driver):  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libgemm8.so gemm_8phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr;

__device__ __forceinline__ half8 lds_read(unsigned addr, int imm) {
  half8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm) : "memory");
  return v;
}

template <int VAR>
__global__ void __launch_bounds__(512, 2) gemm8_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int HT = 16384, KT = 4 * HT;               // half-tile, K-tile buffer; kinds: 0 = A0, 1 = A1, 2 = B0, 3 = B1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware, bijective: XCD x (= block id mod 8) takes the x-th contiguous share of the tile list; tiles n-fastest
  const int gy = N / 256, gx = M / 256, nwg = gx * gy;
  int tile;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bx = tile / gy, by = tile - bx * gy;
  const int m0 = bx * 256, n0 = by * 256;
  const int nt = K / 64;
  constexpr bool NO_A = (VAR & 2) != 0, NO_B = (VAR & 4) != 0;

  // ---- DMA roles: wave w stages row block w (16 rows) of a half-tile, k-halves 0 and 1.  Lane l lands at row l >> 2, 16-byte slot l & 3 of
  // the subtile and therefore FETCHES slot (l & 3) ^ (2 * ((l >> 5) & 1)).
  const int d_row = lane >> 2, d_slot = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  unsigned voff[4];
  {
    const int ra = (wave >> 2) * 128 + (wave & 3) * 16 + d_row;     // + h * 64: tile row of an A half-tile
    const int rb = (wave >> 1) * 64 + (wave & 1) * 16 + d_row;      // + h * 32: tile column of a B half-tile
    voff[0] = (unsigned)(((size_t)(m0 + ra) * K + d_slot * 8) * 2);
    voff[1] = (unsigned)(((size_t)(m0 + ra + 64) * K + d_slot * 8) * 2);
    voff[2] = (unsigned)(((size_t)(n0 + rb) * K + d_slot * 8) * 2);
    voff[3] = (unsigned)(((size_t)(n0 + rb + 32) * K + d_slot * 8) * 2);
  }
  constexpr int SRD = 0x00020000;
  auto stage = [&](auto kindc, int kt) {          // half-tile `kind` of K-tile kt into buffer kt & 1
    constexpr int kind = decltype(kindc)::value;
    char* dst = smem + (kt & 1) * KT + kind * HT + wave * 2048;
    const void* base = kind < 2 ? (const void*)A : (const void*)B;
    const int soff = kt * 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x80000000, SRD), (lptr)dst, 16, (int)voff[kind], soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0x80000000, SRD), (lptr)(dst + 1024), 16, (int)(voff[kind] + 64u), soff, 0, 0);
  };
  auto wait_vm = [&](auto nc) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(nc)::value) : "memory"); };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  // ---- fragment reads: lane l reads row l & 15, 16-byte slot l >> 4 of a subtile (k-step ks = the subtile's k-half)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned fr = lane & 15;
  const unsigned lane_off = (fr * 64 + ((unsigned)lane >> 4) * 16) ^ ((fr >> 3) << 5);
  const unsigned rA = lds0 + wr * 4 * 2048 + lane_off;            // + buffer * KT + h * HT + i * 2048 + ks * 1024
  const unsigned rB = lds0 + 2 * HT + wc * 2 * 2048 + lane_off;

  floatx4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  half8 fa[4][2], fb[2][2];

  auto read_a = [&](unsigned boff, auto hc) {
    constexpr int h = decltype(hc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[i][0] = lds_read(rA + boff, h * HT + i * 2048); fa[i][1] = lds_read(rA + boff, h * HT + i * 2048 + 1024); }
  };
  auto read_b = [&](unsigned boff, auto hc) {
    constexpr int h = decltype(hc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) { fb[i][0] = lds_read(rB + boff, h * HT + i * 2048); fb[i][1] = lds_read(rB + boff, h * HT + i * 2048 + 1024); }
  };
  auto mfma_begin = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (VAR & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!(VAR & 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
  };
  auto mfma_end = [&]() {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto quad = [&](auto ac, auto bc) {          // acc[ah * 4 + i][bh * 2 + j] += A frag i x B frag j
    constexpr int ah = decltype(ac)::value, bh = decltype(bc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][ks], fa[i][ks], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
  };

  // prologue: K-tile 0 complete, K-tile 1 without its B0 (phase 1 of K-tile 0 stages that)
  stage(I0{}, 0); stage(I2{}, 0); stage(I3{}, 0); stage(I1{}, 0);
  if (nt > 1) { if (NO_B) stage(I2{}, 1); stage(I0{}, 1); stage(I3{}, 1); stage(I1{}, 1); wait_vm(std::integral_constant<int, 6>{}); }
  else wait_vm(I0{});
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nt; ++t) {
    const unsigned boff = (t & 1) ? (unsigned)KT : 0u;
    // ---- phase 1: A0 x B0
    read_b(boff, I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_a(boff, I0{});
    if (t + 1 < nt && !NO_B) stage(I2{}, t + 1);
    mfma_begin(); quad(I0{}, I0{}); mfma_end();
    // ---- phase 2: A0 x B1
    read_b(boff, I1{});
    if (t + 2 < nt && !NO_A) stage(I0{}, t + 2);
    mfma_begin(); quad(I0{}, I1{}); mfma_end();
    // ---- phase 3: A1 x B1
    read_a(boff, I1{});
    if (t + 2 < nt && !NO_B) stage(I3{}, t + 2);
    mfma_begin(); quad(I1{}, I1{}); mfma_end();
    // ---- phase 4: A1 x B0 (B0 read again: one fragment buffer per operand)
    read_b(boff, I0{});
    // the wait retires everything K-tile t + 1 reads; what was staged in phases 2-4 of this K-tile stays in flight
    if (t + 2 < nt) { if (!NO_A) stage(I1{}, t + 2); wait_vm(std::integral_constant<int, (NO_A ? 0 : 4) + (NO_B ? 0 : 2)>{}); }
    else wait_vm(I0{});
    mfma_begin(); quad(I1{}, I0{}); mfma_end();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();

  // epilogue: straight from the accumulator layout (lane: row l & 15 of the fragment, 4 consecutive columns 4 * (l >> 4) + j)
  const int fl = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + (i >> 2) * 64 + (i & 3) * 16 + fl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + 4 * fq;
      const half4 h = {(half_t)acc[i][j][0], (half_t)acc[i][j][1], (half_t)acc[i][j][2], (half_t)acc[i][j][3]};
      *reinterpret_cast<half4*>(C + (size_t)m * N + n) = h;
    }
  }
}

template <int VAR>
static int launch(const void* A, const void* B, void* C, int M, int N, int K, hipStream_t s) {
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void*)gemm8_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return -2;
    once = true;
  }
  hipLaunchKernelGGL(gemm8_kernel<VAR>, dim3((M / 256) * (N / 256)), dim3(512), 131072, s, (const half_t*)A, (const half_t*)B, (half_t*)C, M, N, K);
  return (int)hipGetLastError();
}

extern "C" int gemm8_launch(const void* A, const void* B, void* C, int M, int N, int K, int variant, void* stream) {
  if (M % 256 || N % 256 || K % 64 || (size_t)M * K * 2 >= (1ull << 31) || (size_t)N * K * 2 >= (1ull << 31)) return -1;
  hipStream_t s = (hipStream_t)stream;
  switch (variant) {
    case 0: return launch<0>(A, B, C, M, N, K, s);
    case 1: return launch<1>(A, B, C, M, N, K, s);
    case 2: return launch<2>(A, B, C, M, N, K, s);
    case 4: return launch<4>(A, B, C, M, N, K, s);
    case 6: return launch<6>(A, B, C, M, N, K, s);
  }
  return -1;
}

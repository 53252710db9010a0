// Microbenchmark: L2-resident operand fetch ceilings per CU on gfx950:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)      mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: half of the bytes by each path          mode 3: global_load_dwordx4 -> VGPR only (no LDS write)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef float float4_ __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ void __launch_bounds__(256) k(const char* base, size_t ws_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 128-byte row segments, row stride 1280 B (640 fp16 channels): 8 rows x 128 B per wave-instruction
  const int row = lane >> 3, c16 = lane & 7;
  const int rows_per_chunk = U * 4 * 8;
  const size_t nrows = ws_bytes / 1280;
  size_t row0 = ((size_t)blockIdx.x * rows_per_chunk) % (nrows - rows_per_chunk);
  const char* p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = base + (row0 + (size_t)(wave * U + u) * 8 + row) * 1280 + c16 * 16;
  int kofs = 0;
  float4_ acc = {0, 0, 0, 0};
  float4_ regs[U];
  for (int it = 0; it < iters; ++it) {
    char* s = smem + (it % 3) * (U * 4 * 1024);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool dma = MODE == 0 || (MODE == 2 && (u & 1) == 0);
      if (dma) __builtin_amdgcn_global_load_lds((gptr)(p[u] + kofs), (lptr)(s + (wave * U + u) * 1024), 16, 0, 0);
      else regs[u] = __builtin_nontemporal_load(reinterpret_cast<const float4_*>(p[u] + kofs));
    }
    kofs += 128;
    if (kofs >= 1280) {
      kofs = 0;
      row0 = (row0 + (size_t)gridDim.x * rows_per_chunk) % (nrows - rows_per_chunk);
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = base + (row0 + (size_t)(wave * U + u) * 8 + row) * 1280 + c16 * 16;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool dma = MODE == 0 || (MODE == 2 && (u & 1) == 0);
      if (!dma) {
        if (MODE == 3) acc += regs[u];
        else *reinterpret_cast<float4_*>(s + (wave * U + u) * 1024 + lane * 16) = regs[u];
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && tid == 0) sink[blockIdx.x] = ((float*)smem)[0] + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE, int U>
void run(const char* buf, size_t ws, int bpc, float* sink, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 400, blocks = 256 * bpc;
  size_t dyn = (160 * 1024) / bpc - 1024;
  if (dyn < 3 * U * 4 * 1024) dyn = 3 * U * 4 * 1024;
  hipFuncSetAttribute((const void*)k<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  k<MODE, U><<<blocks, 256, dyn>>>(buf, ws, 20, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, U><<<blocks, 256, dyn>>>(buf, ws, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double tbs = (double)blocks * iters * U * 4 * 1024 / (ms * 1e-3) / 1e12;
  printf("mode %d  ws %5zu MB  bpc %d  U %d | %6.2f TB/s  %5.1f B/clk/CU %s\n", MODE, ws >> 20, bpc, U, tbs, tbs * 1e12 / 256 / 2.4e9,
         hipGetLastError() == hipSuccess ? "" : "ERR");
}

int main() {
  size_t cap = (size_t)1 << 30;
  char* buf; (void)hipMalloc(&buf, cap); (void)hipMemset(buf, 0, cap);
  float* sink; (void)hipMalloc(&sink, 1 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t ws : {(size_t)1 << 20, (size_t)3 << 20})
    for (int bpc : {1, 2, 4}) {
      run<0, 4>(buf, ws, bpc, sink, e0, e1);
      run<1, 4>(buf, ws, bpc, sink, e0, e1);
      run<2, 4>(buf, ws, bpc, sink, e0, e1);
      run<3, 4>(buf, ws, bpc, sink, e0, e1);
      run<3, 8>(buf, ws, bpc, sink, e0, e1);
    }
  return 0;
}

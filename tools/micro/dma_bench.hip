// Microbenchmark: achievable global -> LDS (global_load_lds_dwordx4) bandwidth per CU on gfx950 as a function of the
// contiguous segment each row contributes (64 B ... 1 KiB), the row stride, the working-set size and blocks per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

template <int NST, int U>
__global__ void __launch_bounds__(256) dma_kernel(const char* base, size_t ws_bytes, int seg, int stride, int row_len, int iters, int* sink, int share) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lps = seg / 16;                  // lanes per row segment
  const int rpi = 64 / lps;                  // rows per instruction
  const int row = lane / lps, c16 = lane % lps;
  // block's tile: U*4 instructions per chunk => U*4*rpi rows; tiles laid consecutively in a [rows][row_len] matrix of given stride
  const int rows_per_chunk = U * 4 * rpi;
  const size_t nrows_total = ws_bytes / stride;
  // share = S: groups of S consecutive blocks read identical addresses at the same time (operand sharing as in a GEMM)
  const size_t bgrp = blockIdx.x / share, ngrp = (gridDim.x + share - 1) / share;
  size_t row0 = (bgrp * rows_per_chunk) % (nrows_total - rows_per_chunk);
  const char* p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = base + (row0 + (size_t)(wave * U + u) * rpi + row) * stride + c16 * 16;
  int kofs = 0;
  for (int it = 0; it < iters; ++it) {
    char* s = smem + (it % NST) * (U * 4 * 1024);
#pragma unroll
    for (int u = 0; u < U; ++u)
      __builtin_amdgcn_global_load_lds((gptr)(p[u] + kofs), (lptr)(s + (wave * U + u) * 1024), 16, 0, 0);
    kofs += seg;
    if (kofs >= row_len) {   // next band of rows (stay inside the working set)
      kofs = 0;
      row0 = (row0 + ngrp * rows_per_chunk) % (nrows_total - rows_per_chunk);
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = base + (row0 + (size_t)(wave * U + u) * rpi + row) * stride + c16 * 16;
    }
    if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (U == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (U == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && tid == 0) sink[blockIdx.x] = ((int*)smem)[0];
}

int main(int argc, char** argv) {
  size_t cap = (size_t)1 << 30;
  char* buf; hipMalloc(&buf, cap); hipMemset(buf, 1, cap);
  int* sink; hipMalloc(&sink, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%6s %7s %8s %4s %3s %6s | %9s %9s\n", "seg", "stride", "ws_MB", "bpc", "U", "nst", "TB/s", "B/clk/CU");
  const int segs[] = {64, 128};
  const size_t wss[] = {(size_t)2 << 20, (size_t)24 << 20};
  for (int share : {1, 4, 8, 32, 128})
  for (size_t ws : wss)
    for (int seg : segs)
      for (int bpc = 2; bpc <= 2; bpc *= 2)
        for (int U = 4; U <= 4; U += 2)
          for (int nst = 3; nst <= 3; ++nst) {
            int stride = seg == 1024 ? 1024 : 640 * 2;       // 640 channels fp16 (or fully contiguous)
            int row_len = seg == 1024 ? 1024 : 1280;
            int iters = 400;
            int blocks = 256 * bpc;
            size_t lds = (size_t)nst * U * 4 * 1024;
            // pad LDS so that exactly bpc blocks fit per CU (160 KB)
            size_t want = (160 * 1024) / bpc; if (want > 64 * 1024 && bpc > 1) want = 160 * 1024 / bpc;
            size_t dyn = want - 1024 > lds ? want - 1024 : lds;
            if (dyn > 160 * 1024 - 512) dyn = 160 * 1024 - 512;
            auto kern = U == 2 ? dma_kernel<3, 2> : dma_kernel<3, 4>;
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            kern<<<blocks, 256, dyn>>>(buf, ws, seg, stride, row_len, 20, sink, share);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            kern<<<blocks, 256, dyn>>>(buf, ws, seg, stride, row_len, iters, sink, share);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double bytes = (double)blocks * iters * U * 4 * 1024;
            double tbs = bytes / (ms * 1e-3) / 1e12;
            printf("share %3d %6d %7d %8zu %4d %3d %6d | %9.2f %9.1f   %s\n", share, seg, stride, ws >> 20, bpc, U, nst, tbs, tbs * 1e12 / 256 / 2.4e9,
                   hipGetLastError() == hipSuccess ? "" : "ERR");
          }
  return 0;
}

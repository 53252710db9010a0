// Microbenchmark: the global -> LDS DMA rate of a CU against how the requests are issued -- NW waves (4 or 8) each keeping up to
// 2 x DEPTH buffer_load_dwordx4 ... lds instructions (1 KB each: 8 rows x 128 bytes, row stride 23040 bytes) in flight, no barriers -- and
// against where the lines come from:
//   shared : every CU walks the same 12 MB (all hits in L2 / MALL after the first pass)
//   panels : the sharing of a 12288 x 1280 x 11520 GEMM on 256 x 256 tiles in the kernel's XCD-aware order: CU i of XCD x (i < 30) reads the
//            "activation" panel x * 6 + i / 5 (shared with the 4 other n-tiles of its m-panel, same XCD) for one half of its instructions and
//            the "weight" panel i % 5 (shared with every m-tile, all XCDs) for the other half; 48 + 5 panels of 5.9 MB
// The ping-pong kernel issues from 8 waves with 8 (ring experiment: 12) instructions in flight per wave = 64 (96) KB per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_depth_sweep dma_depth_sweep.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lptr;

template <int NW, int DEPTH>
__global__ void __launch_bounds__(512, 2) sweep_kernel(const char* base, size_t panel_bytes, int mode, int steps, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= NW) return;
  const size_t stride = 23040;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  if (mode >= 1 && idx >= 30) return;
  const char* srcA = base, *srcW = base;
  if (mode >= 1) { srcA = base + (size_t)(xcd * 6 + idx / 5) * panel_bytes; srcW = base + (size_t)(48 + idx % 5) * panel_bytes; }
  if (mode == 2 || mode == 5) srcW = srcA + 128 * stride * 0;   // activation panels only (both halves of a K-tile from the m-panel: rows 0-255 twice)
  if (mode == 3) srcA = srcW;                        // weight panels only
  const unsigned rot = mode == 4 ? (unsigned)(idx % 5 + idx / 5) : 0u;   // the sharers of a panel one K-tile apart
  unsigned lane_off = (unsigned)((lane >> 3) * stride + (lane & 7) * 16);
  if (mode == 7) lane_off = (unsigned)((lane >> 2) * stride + (lane & 3) * 16);      // 64-byte row segments (a K-tile of 32): 16 rows per instruction
  // a step = DEPTH instructions of this wave; the waves of a CU cover rows [wave * DEPTH * 8, ...) of the 256-row panel slab, wrapping
  char* ring = smem + wave * (2 * DEPTH * 1024);
  for (int s = 0; s < steps; ++s) {
    char* dst = ring + (s & 1) * DEPTH * 1024;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const unsigned blk = (unsigned)((s * NW + wave) * DEPTH + i);
      // 64 instructions = one K-tile: the 256 activation rows, then the 256 weight rows, at k position blk / 64 (180 positions, each line once)
      unsigned row = (blk * 8) & 255, kofs = (((blk >> 6) + rot) % 180) * 128;
      if (mode == 7) { row = (blk * 16) & 255; kofs = ((blk >> 5) % 360) * 64; }        // 32 instructions = one K-tile of 32: 256 + 256 rows x 64 bytes
      const char* src = (mode == 7 ? (blk & 16) : (blk & 32)) ? srcW : srcA;
      unsigned off = row * (unsigned)stride + kofs + lane_off;
      if (mode >= 5 && (mode == 5 || !(blk & 32))) off = (kofs >> 7) * 32768u + row * 128u + (unsigned)lane * 16u;     // K-tile-blocked panel: a K-tile's 256 rows x 128 bytes are contiguous
      __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x80000000, 0x00020000), (lptr)(dst + i * 1024), 16, (int)off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lane == 0 && ((int*)ring)[0] == 0x7fffffff) sink[blockIdx.x] = 1;
}

// The ping-pong kernel's DMA-only form: 8 waves, per phase every wave issues 2 instructions, waits until at most KEEP remain in flight
// (8 = four batches, the kernel's schedule) and passes NBAR workgroup barriers.
template <int KEEP, int NBAR>
__global__ void __launch_bounds__(512, 2) pp_like_kernel(const char* base, size_t panel_bytes, int mode, int steps, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t stride = 23040;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  if (mode >= 1 && idx >= 30) return;
  const char* srcA = base, *srcW = base;
  if (mode >= 1) { srcA = base + (size_t)(xcd * 6 + idx / 5) * panel_bytes; srcW = base + (size_t)(48 + idx % 5) * panel_bytes; }
  if (mode == 2 || mode == 5) srcW = srcA + 128 * stride * 0;   // activation panels only (both halves of a K-tile from the m-panel: rows 0-255 twice)
  if (mode == 3) srcA = srcW;                        // weight panels only
  const unsigned rot = mode == 4 ? (unsigned)(idx % 5 + idx / 5) : 0u;   // the sharers of a panel one K-tile apart
  const unsigned lane_off = (unsigned)((lane >> 3) * stride + (lane & 7) * 16);
  for (int s = 0; s < steps; ++s) {
    char* dst = smem + ((s & 7) * 8 + wave) * 2048;            // ring of 8 batches x 16 KB
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned blk = (unsigned)((s * 8 + wave) * 2 + i);
      // 64 instructions = one K-tile: the 256 activation rows, then the 256 weight rows, at k position blk / 64 (180 positions, each line once)
      unsigned row = (blk * 8) & 255, kofs = (((blk >> 6) + rot) % 180) * 128;
      if (mode == 7) { row = (blk * 16) & 255; kofs = ((blk >> 5) % 360) * 64; }        // 32 instructions = one K-tile of 32: 256 + 256 rows x 64 bytes
      const char* src = (mode == 7 ? (blk & 16) : (blk & 32)) ? srcW : srcA;
      unsigned off = row * (unsigned)stride + kofs + lane_off;
      if (mode >= 5 && (mode == 5 || !(blk & 32))) off = (kofs >> 7) * 32768u + row * 128u + (unsigned)lane * 16u;     // K-tile-blocked panel: a K-tile's 256 rows x 128 bytes are contiguous
      __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x80000000, 0x00020000), (lptr)(dst + i * 1024), 16, (int)off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
#pragma unroll
    for (int b = 0; b < NBAR; ++b) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lane == 0 && ((int*)smem)[wave * 512] == 0x7fffffff) sink[blockIdx.x] = 1;
}

template <int KEEP, int NBAR>
static void run_pp(const char* buf, size_t panel_bytes, int mode, int* sink) {
  const int lds = 128 * 1024;
  hipFuncSetAttribute((const void*)pp_like_kernel<KEEP, NBAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 11520 / 16;
  pp_like_kernel<KEEP, NBAR><<<256, 512, lds>>>(buf, panel_bytes, mode, steps / 8, sink);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    pp_like_kernel<KEEP, NBAR><<<256, 512, lds>>>(buf, panel_bytes, mode, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-7s pp-like: 2 instructions per wave and phase, vmcnt(%2d), %d barriers per phase  %8.1f us  %6.1f B/clk/CU   %s\n", mode == 0 ? "shared" : mode == 1 ? "panels" : mode == 2 ? "A only" : mode == 3 ? "W only" : mode == 4 ? "rotated" : mode == 5 ? "A blk" : mode == 6 ? "pan blk" : "pan 64B", KEEP, NBAR, best * 1e3,
         (double)steps * 16 * 1024 / (best * 1e-3) / 2.4e9, hipGetLastError() == hipSuccess ? "" : "ERR");
}

template <int NW, int DEPTH>
static void run(const char* buf, size_t panel_bytes, int mode, int* sink) {
  const int lds = NW * 2 * DEPTH * 1024 > 96 * 1024 ? NW * 2 * DEPTH * 1024 : 96 * 1024;     // >= 96 KB: one workgroup per CU
  hipFuncSetAttribute((const void*)sweep_kernel<NW, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int kb_per_cu = 11520;                                  // the KB a CU moves (180 K-tiles of 64 KB)
  const int steps = kb_per_cu / (NW * DEPTH);
  sweep_kernel<NW, DEPTH><<<256, 512, lds>>>(buf, panel_bytes, mode, steps / 8, sink);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    sweep_kernel<NW, DEPTH><<<256, 512, lds>>>(buf, panel_bytes, mode, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (double)steps * NW * DEPTH * 1024;
  printf("%-7s waves %d  in flight <= %3d KB/CU  %8.1f us  %6.1f B/clk/CU at 2.4 GHz   %s\n", mode == 0 ? "shared" : mode == 1 ? "panels" : mode == 2 ? "A only" : mode == 3 ? "W only" : mode == 4 ? "rotated" : mode == 5 ? "A blk" : mode == 6 ? "pan blk" : "pan 64B", NW, NW * 2 * DEPTH, best * 1e3, bytes / (best * 1e-3) / 2.4e9,
         hipGetLastError() == hipSuccess ? "" : "ERR");
}

int main() {
  const size_t panel = (size_t)256 * 23040;                     // 5.9 MB
  char* buf; if (hipMalloc(&buf, panel * 53 + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, panel * 53 + (1 << 20));
  int* sink; hipMalloc(&sink, 4096);
  const char* names[8] = {"shared", "panels", "A only", "W only", "rotated", "A only, K-tile-blocked layout", "panels, A K-tile-blocked", "panels, 64-byte row segments (K-tile 32)"};
  for (int mode = 0; mode <= 7; ++mode) {
    printf("--- source: %s\n", names[mode]);
    if (mode <= 1) { run<4, 2>(buf, panel, mode, sink); run<4, 4>(buf, panel, mode, sink); run<4, 8>(buf, panel, mode, sink); run<4, 16>(buf, panel, mode, sink); run<8, 2>(buf, panel, mode, sink); }
    run<8, 4>(buf, panel, mode, sink); run<8, 8>(buf, panel, mode, sink);
    run_pp<8, 2>(buf, panel, mode, sink);
    if (mode <= 1) { run_pp<8, 0>(buf, panel, mode, sink); run_pp<12, 2>(buf, panel, mode, sink); run_pp<4, 2>(buf, panel, mode, sink); }
  }
  return 0;
}

// One instantiation of the LDS-DMA implicit-GEMM kernel, for reading its ISA:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c tools/isa_probe.hip -I pnpinversion_amd/csrc -DPROBE_ARGS="128,128,64,2,2,0,1"
#include "igemm_dma.inc"
#ifndef PROBE_ARGS
#define PROBE_ARGS 128, 128, 64, 2, 2, 0, 1
#endif
template __global__ void igemm_dma_kernel<PROBE_ARGS>(GemmP, const half_t*);

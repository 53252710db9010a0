// One instantiation of the 8-wave ping-pong implicit-GEMM kernel, for reading its ISA:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c tools/isa_probe_pp.hip -I pnpinversion_amd/csrc -DPROBE_ARGS="256,256,2,4"
#include <type_traits>
#include "igemm_pp.inc"
#ifndef PROBE_ARGS
#define PROBE_ARGS 256, 256, 2, 4
#endif
template __global__ void igemm_pp_kernel<PROBE_ARGS>(GemmP);

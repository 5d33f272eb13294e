// The plain sweep of small meshes with the commonest switches compiled in (k_sweep<1, false, false, false, false, LREC, FAST>,
// nk_kernels.h) as a translation unit of its own: it is compiled WITH the machine-level loop-invariant code motion that the
// rest of the library turns off (Makefile).  LICM keeps the polynomials' FP64 constants and the table addresses in registers
// across the tile loop instead of rebuilding them with v_mov pairs in every iteration: 156 instead of 126 VGPRs -- still
// three workgroups per CU -- and 5.8 % fewer VALU instructions per launch (SQ_INSTS_VALU 67.3M against 71.4M at 1e7
// particles), k_sweep 0.1898 / 0.1917 against 0.1967 / 0.1956 ms (profiles/r03_notes.txt (22)).  The instantiations that LICM
// pushes over a residency step (rough facets: 197 against 166 VGPRs) stay in nk_engine.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nanokappa_hip.h"
#define NK_KERNEL_LINKAGE static        // (only the template instantiations below leave this file)
#include "nk_kernels.h"

template __global__ void k_sweep<1, false, false, false, false, true, 1>(NkDev, uint32_t, int, int);
template __global__ void k_sweep<1, false, false, false, false, true, 2>(NkDev, uint32_t, int, int);
template __global__ void k_sweep<1, false, false, false, false, false, 1>(NkDev, uint32_t, int, int);
template __global__ void k_sweep<1, false, false, false, false, false, 2>(NkDev, uint32_t, int, int);
// ... and the same four for the box store (BOX: no cached next hit in the particle state, nk_kernels.h)
template __global__ void k_sweep<1, false, false, false, false, true, 1, true>(NkDev, uint32_t, int, int);
template __global__ void k_sweep<1, false, false, false, false, true, 2, true>(NkDev, uint32_t, int, int);
template __global__ void k_sweep<1, false, false, false, false, false, 1, true>(NkDev, uint32_t, int, int);
template __global__ void k_sweep<1, false, false, false, false, false, 2, true>(NkDev, uint32_t, int, int);

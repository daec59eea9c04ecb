// nmx_timeosc.hip -- the time / oscillatory and Hilbert kernels at their default launch width, compiled with
// -DNMX_BLOCK_FIXED=128: NMX_NT is a constant, so the grid-stride loops of the statically planned
// transforms and of the fused scans have compile-time trip counts.  nmx_api.hip keeps the
// run-time-width version for other widths (NMX_NT_TIMEOSC, windows > 1024 samples).
#ifndef NMX_BLOCK_FIXED
#error "compile with -DNMX_BLOCK_FIXED=128"
#endif
#include <hip/hip_runtime.h>

#define NMX_CAT2(a, b) a##b
#define NMX_CAT(a, b) NMX_CAT2(a, b)
#define NMX_STR2(a) #a
#define NMX_STR(a) NMX_STR2(a)

#include "nmx_k_bank_w64.h"
#include "nmx_k_timeosc.h"

extern __shared__ __attribute__((aligned(16))) float nmx_smem_to[];

__global__ void __launch_bounds__(NMX_BLOCK_FIXED) NMX_CAT(nmx_kern_timeosc_fixed, NMX_BLOCK_FIXED)(const NmxTimeOscArgs A) {
  const int item = blockIdx.x;
  nmx_time_osc_item(A, item / A.n_channels, item % A.n_channels, nmx_smem_to);
}

__global__ void __launch_bounds__(NMX_BLOCK_FIXED) NMX_CAT(nmx_kern_hilbert_fixed, NMX_BLOCK_FIXED)(const NmxHilbertArgs A) {
  nmx_hilbert_item(A, (long long)blockIdx.x, nmx_smem_to);
}


extern "C" void NMX_CAT(nmx_hilbert_fixed_launch, NMX_BLOCK_FIXED)(const NmxHilbertArgs* A, long long n_items, size_t lds, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_hilbert_fixed, NMX_BLOCK_FIXED), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  hipLaunchKernelGGL(NMX_CAT(nmx_kern_hilbert_fixed, NMX_BLOCK_FIXED), dim3((unsigned)n_items), dim3(NMX_BLOCK_FIXED), lds, s, *A);
  nmxi_note_kernel("nmx_kern_hilbert_fixed" NMX_STR(NMX_BLOCK_FIXED));
}

extern "C" void NMX_CAT(nmx_timeosc_fixed_launch, NMX_BLOCK_FIXED)(const NmxTimeOscArgs* A, int n_items, size_t lds, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_timeosc_fixed, NMX_BLOCK_FIXED), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  hipLaunchKernelGGL(NMX_CAT(nmx_kern_timeosc_fixed, NMX_BLOCK_FIXED), dim3(n_items), dim3(NMX_BLOCK_FIXED), lds, s, *A);
  nmxi_note_kernel("nmx_kern_timeosc_fixed" NMX_STR(NMX_BLOCK_FIXED));
}

// nmx_w64.hip -- translation unit of the single-wave FIR-bank kernel (nmx_k_bank_w64.h).
// Built twice by __graft_entry__.build_lib(): default flags (-> *_slp: clang's SLP vectoriser
// packs the complex arithmetic into v_pk_*_f32) and -fno-slp-vectorize (-> *_scalar: no
// packing, fewer register-pair moves, no scratch spills).  libnmx picks one at run time
// (NMX_W64_VARIANT, default chosen from measurements -- see DESIGN.md).
#include <hip/hip_runtime.h>

#include "nmx_k_bank_w64.h"

#ifndef NMX_W64_NAME
#error "define NMX_W64_NAME"
#endif
#define NMX_CAT2(a, b) a##b
#define NMX_CAT(a, b) NMX_CAT2(a, b)

extern __shared__ __attribute__((aligned(16))) float nmx_smem_w64[];

// register budgets: the FIR-bank instantiation fits 168 VGPRs (3 waves/SIMD, 12 per CU, 5 spilled
// dwords); the notch instantiation (odd-reflection staging) needs the 256-VGPR budget.
__global__ void __launch_bounds__(64, 3) NMX_CAT(nmx_kern_bank_w64_, NMX_W64_NAME)(const NmxBankW64Args A) {
  const int item = blockIdx.x;
  nmx_bank_w64_item<0>(A, item / A.b.n_channels, item % A.b.n_channels, nmx_smem_w64);
}
__global__ void __launch_bounds__(64, 2) NMX_CAT(nmx_kern_notch_w64_, NMX_W64_NAME)(const NmxBankW64Args A) {
  const int item = blockIdx.x;
  nmx_bank_w64_item<1>(A, item / A.b.n_channels, item % A.b.n_channels, nmx_smem_w64);
}

extern "C" void NMX_CAT(nmx_w64_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, size_t lds,
                                                       hipStream_t s) {
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64_, NMX_W64_NAME),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_notch_w64_, NMX_W64_NAME),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (A->b.pad_mode == 0)
    hipLaunchKernelGGL(NMX_CAT(nmx_kern_bank_w64_, NMX_W64_NAME), dim3(n_items), dim3(64), lds, s, *A);
  else
    hipLaunchKernelGGL(NMX_CAT(nmx_kern_notch_w64_, NMX_W64_NAME), dim3(n_items), dim3(64), lds, s, *A);
}

// nmx_specmm.hip -- translation unit of the matrix-pipe spectrum kernel (nmx_k_specmm.h): FFT band power + Hjorth /
// LineLength / Raw of 1000-sample windows, 32 windows per wave, four waves per workgroup (they share the staged DFT table).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "nmx_k_specmm.h"

extern __shared__ __attribute__((aligned(16))) float nmx_smem_smm[];

template <int NB, bool TD, bool CLEAN>
__global__ void __launch_bounds__(256, 2) nmx_kern_specmm_w1000(const NmxTimeOscArgs A0, long long n_items) {
  // (kernel-argument pointer laundered once per tile: the plan -- 150 dwords -- is re-read with s_load instead of being
  // hoisted, together with every loop-invariant band mask, into scalar registers that spill)
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  const long long n_tiles = (n_items + 127) / 128;
#pragma unroll 1
  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    asm volatile("" : "+s"(Ap));
    nmx_specmm_tile<NB, TD, CLEAN>(*(const NmxTimeOscArgs*)Ap, 128 * t, n_items, nmx_smem_smm);
  }
}

// returns 0 when the configuration needs another kernel
extern "C" int nmx_specmm_launch(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  static int on = -1;
  if (on < 0) { const char* v = getenv("NMX_SPECMM"); on = !(v && v[0] == '0'); }
  if (!on || !nmx_specmm_ok(*A) || n_items < 1) return 0;
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long long n_tiles = ((long long)n_items + 127) / 128;
  static int wg_per_cu = 0;
  if (!wg_per_cu) { const char* v = getenv("NMX_SPECMM_WG"); wg_per_cu = (v && atoi(v) >= 1 && atoi(v) <= 4) ? atoi(v) : 2; }
  long long grid = (long long)n_cu * wg_per_cu;
  if (grid > n_tiles) grid = n_tiles;
  const size_t lds = (size_t)NMX_SMM_LDS_FLOATS * 4;
  const bool td = (A->features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) != 0, clean = A->clean_on_load != 0;
#define NMX_SMM_LAUNCH(NB, TD, CL)                                                                                   \
  hipLaunchKernelGGL((nmx_kern_specmm_w1000<NB, TD, CL>), dim3((unsigned)grid), dim3(256), lds, s, *A, (long long)n_items)
  if (A->n_bands <= 4) {
    if (td && clean) NMX_SMM_LAUNCH(4, true, true); else if (td) NMX_SMM_LAUNCH(4, true, false);
    else if (clean) NMX_SMM_LAUNCH(4, false, true); else NMX_SMM_LAUNCH(4, false, false);
    nmxi_note_kernel("nmx_kern_specmm_w1000<4>");
  } else {
    if (td && clean) NMX_SMM_LAUNCH(8, true, true); else if (td) NMX_SMM_LAUNCH(8, true, false);
    else if (clean) NMX_SMM_LAUNCH(8, false, true); else NMX_SMM_LAUNCH(8, false, false);
    nmxi_note_kernel("nmx_kern_specmm_w1000<8>");
  }
#undef NMX_SMM_LAUNCH
  return 1;
}

// nmx_specmm.hip -- translation unit of the matrix-pipe spectrum kernel (nmx_k_specmm.h): FFT band power + Hjorth /
// LineLength / Raw of 1000-sample windows, 32 windows per wave, four waves per workgroup (they share the staged DFT table).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "nmx_k_specmm.h"

extern __shared__ __attribute__((aligned(16))) float nmx_smem_smm[];

// RING = 3: two workgroups per CU (256 registers per lane); RING = 4: one (512)
template <int NB, bool TD, bool CLEAN, int RING>
__global__ void __launch_bounds__(256, RING == 4 ? 1 : 2) nmx_kern_specmm_w1000(const NmxTimeOscArgs A0, long long n_items) {
  // (kernel-argument pointer laundered once per tile: the plan -- 150 dwords -- is re-read with s_load instead of being
  // hoisted, together with every loop-invariant band mask, into scalar registers that spill)
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  const int C = ((const NmxTimeOscArgs*)Ap)->n_channels, n_windows = (int)(n_items / C);
  const long long n_groups = (long long)((n_windows + 31) / 32) * C, n_tiles = (n_groups + 3) / 4;
#pragma unroll 1
  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    asm volatile("" : "+s"(Ap));
    nmx_specmm_tile<NB, TD, CLEAN, RING>(*(const NmxTimeOscArgs*)Ap, 4 * t, n_windows, nmx_smem_smm);
  }
}

// returns 0 when the configuration needs another kernel
extern "C" int nmx_specmm_launch(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  static int on = -1;
  // (opt-in while it is slower than nmx_kern_timeosc_w1000_low on the Mode A workload: 1.84 ms against 1.40 per 1 M windows;
  // the same launch with cache-friendly loads 1.20 ms, the matrix pipe's own floor 0.86 -- profiles/r04_specmm.txt)
  if (on < 0) { const char* v = getenv("NMX_SPECMM"); on = (v && v[0] == '1') ? 1 : 0; }
  if (!on || !nmx_specmm_ok(*A) || n_items < 1) return 0;
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int n_win = n_items / A->n_channels;
  const long long n_tiles = ((long long)((n_win + 31) / 32) * A->n_channels + 3) / 4;
  static int wg_per_cu = 0, ring = 0;
  if (!wg_per_cu) { const char* v = getenv("NMX_SPECMM_WG"); wg_per_cu = (v && atoi(v) >= 1 && atoi(v) <= 4) ? atoi(v) : 2; }
  if (!ring) { const char* v = getenv("NMX_SPECMM_RING"); ring = (v && atoi(v) == 4) ? 4 : 3; }
  if (ring == 4) wg_per_cu = 1;
  long long grid = (long long)n_cu * wg_per_cu;
  if (grid > n_tiles) grid = n_tiles;
  const size_t lds = (size_t)NMX_SMM_LDS_FLOATS * 4;
  const bool td = (A->features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) != 0, clean = A->clean_on_load != 0;
#define NMX_SMM_LAUNCH(NB, TD, CL)                                                                                          \
  do {                                                                                                                      \
    if (ring == 4) hipLaunchKernelGGL((nmx_kern_specmm_w1000<NB, TD, CL, 4>), dim3((unsigned)grid), dim3(256), lds, s, *A, (long long)n_items); \
    else hipLaunchKernelGGL((nmx_kern_specmm_w1000<NB, TD, CL, 3>), dim3((unsigned)grid), dim3(256), lds, s, *A, (long long)n_items);          \
  } while (0)
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

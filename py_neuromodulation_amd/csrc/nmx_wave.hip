// nmx_wave.hip -- translation unit of the one-item-per-WAVE kernels (burst statistics, sharp-wave
// analysis).  Compiled with -DNMX_NT_FIXED=64: inside an item NMX_TID is the lane, NMX_NT is the
// constant 64, NMX_SYNC() is a wave-local LDS fence and the block reductions are shuffle-only.
// A workgroup carries `k` independent waves (own LDS slice each), so there is no workgroup barrier
// anywhere in these kernels.
#ifndef NMX_NT_FIXED
#error "compile with -DNMX_NT_FIXED=64"
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include <cstdlib>

#include "nmx_k_bursts.h"
#include "nmx_k_bank_w64.h"
#include "nmx_k_scan.h"
#include "nmx_k_timeosc_w1000.h"
#include "nmx_k_timeosc_w510.h"
#include "nmx_k_sharpwave.h"

extern __shared__ __attribute__((aligned(16))) float nmx_smem_wave[];

__global__ void __launch_bounds__(256) nmx_kern_burst_stat(const NmxBurstStatArgs A, int n_items, int slice) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPR
  const int item = blockIdx.x * (blockDim.x >> 6) + wave;
  if (item >= n_items) return;
  const int bi = item % A.n_bands, r = item / A.n_bands;
  nmx_burst_stat_item(A, r / A.n_channels, r % A.n_channels, bi, nmx_smem_wave + wave * slice);
}

__global__ void __launch_bounds__(256) nmx_kern_sharp(const NmxSharpArgs A, int n_items, int slice) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPR
  const int item = blockIdx.x * (blockDim.x >> 6) + wave;
  if (item >= n_items) return;
  const int fi = item % A.n_filters, r = item / A.n_filters;
  nmx_sharp_item(A, r / A.n_channels, r % A.n_channels, fi, nmx_smem_wave + wave * slice);
}

// threshold walk, steady regime: one wave per (channel, band); NR = registers per lane for a hop's new samples
template <int NR>
__global__ void __launch_bounds__(64) nmx_kern_burst_thr_wave(const NmxBurstThrArgs A) {
  __builtin_amdgcn_s_setprio(3);   // sequential and on the critical path: win issue arbitration
  const int item = blockIdx.x;
  nmx_burst_thr_wave_item<NR>(A, item / A.n_bands, item % A.n_bands, nmx_smem_wave);
}

extern "C" void nmx_wave_launch_burst_thr(const NmxBurstThrArgs* A, int n_items, hipStream_t s) {
  if (A->overlap <= 128) {
    const size_t lds = (size_t)(NMX_THRW_LDS_FLOATS_NR(2) + A->K / 64 + 4) * 4;
    hipLaunchKernelGGL(nmx_kern_burst_thr_wave<2>, dim3(n_items), dim3(64), lds, s, *A);
    nmxi_note_kernel("nmx_kern_burst_thr_wave<2>");
  } else {
    static unsigned long long seen = 0;
    if (nmx_first_on_device(seen))
      (void)hipFuncSetAttribute((const void*)nmx_kern_burst_thr_wave<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = (size_t)(NMX_THRW_LDS_FLOATS_NR(4) + A->K / 64 + 4) * 4;
    hipLaunchKernelGGL(nmx_kern_burst_thr_wave<4>, dim3(n_items), dim3(64), lds, s, *A);
    nmxi_note_kernel("nmx_kern_burst_thr_wave<4>");
  }
}

// Hilbert envelope of length-1000 series, one wave per series (wave-level 500-point transforms)
__global__ void __launch_bounds__(64) nmx_kern_hilbert_w500(const NmxHilbertArgs A) {
  nmx_hilbert_w500_item(A, (long long)blockIdx.x, nmx_smem_wave);
}

extern "C" void nmx_wave_launch_hilbert_w500(const NmxHilbertArgs* A, long long n_items, hipStream_t s) {
  hipLaunchKernelGGL(nmx_kern_hilbert_w500, dim3((unsigned)n_items), dim3(64), (size_t)NMX_W500_LDS_FLOATS * 4, s, *A);
  nmxi_note_kernel("nmx_kern_hilbert_w500");
}

// time-domain + FFT / Welch / STFT band means of the default shape, one wave per (window, channel)
template <int NB>
__global__ void __launch_bounds__(64) nmx_kern_timeosc_w1000(const NmxTimeOscArgs A) {
  const int item = blockIdx.x;
  nmx_timeosc_w1000_item<NB>(A, item / A.n_channels, item % A.n_channels, nmx_smem_wave);
}

// returns 0 when the configuration needs the generic kernel
extern "C" int nmx_wave_launch_timeosc_w1000(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  if (!nmx_timeosc_w1000_ok(*A)) return 0;
  if (A->n_bands <= 4) {
    hipLaunchKernelGGL(nmx_kern_timeosc_w1000<4>, dim3(n_items), dim3(64), (size_t)NMX_TOW_LDS_FLOATS * 4, s, *A);
    nmxi_note_kernel("nmx_kern_timeosc_w1000<4>");
  } else {
    hipLaunchKernelGGL(nmx_kern_timeosc_w1000<8>, dim3(n_items), dim3(64), (size_t)NMX_TOW_LDS_FLOATS * 4, s, *A);
    nmxi_note_kernel("nmx_kern_timeosc_w1000<8>");
  }
  return 1;
}

// STFT with 500-sample segments on windows of other lengths (<= 2048), the only time / oscillatory feature
template <int NB>
__global__ void __launch_bounds__(64) nmx_kern_timeosc_stft500(const NmxTimeOscArgs A) {
  const int item = blockIdx.x;
  nmx_timeosc_stft500_item<NB>(A, item / A.n_channels, item % A.n_channels, nmx_smem_wave);
}
extern "C" int nmx_wave_launch_timeosc_stft500(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  if (!nmx_timeosc_stft500_ok(*A)) return 0;
  if (A->n_bands <= 4) {
    hipLaunchKernelGGL(nmx_kern_timeosc_stft500<4>, dim3(n_items), dim3(64), (size_t)NMX_TOS_LDS_FLOATS * 4, s, *A);
    nmxi_note_kernel("nmx_kern_timeosc_stft500<4>");
  } else {
    hipLaunchKernelGGL(nmx_kern_timeosc_stft500<8>, dim3(n_items), dim3(64), (size_t)NMX_TOS_LDS_FLOATS * 4, s, *A);
    nmxi_note_kernel("nmx_kern_timeosc_stft500<8>");
  }
  return 1;
}

// 510-sample transforms (17 ms at 30 kHz): prime-factor transform per wave (nmx_k_timeosc_w510.h)
template <int NB>
__global__ void __launch_bounds__(64) nmx_kern_timeosc_w510(const NmxTimeOscArgs A) {
  const int item = blockIdx.x;
  nmx_timeosc_w510_item<NB>(A, A.w510_tab, item / A.n_channels, item % A.n_channels, nmx_smem_wave);
}
extern "C" int nmx_wave_launch_timeosc_w510(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  if (!nmx_timeosc_w510_ok(*A, A->w510_tab)) return 0;
  if (A->n_bands <= 4) {
    hipLaunchKernelGGL(nmx_kern_timeosc_w510<4>, dim3(n_items), dim3(64), (size_t)NMX_TO510_LDS_FLOATS * 4, s, *A);
    nmxi_note_kernel("nmx_kern_timeosc_w510<4>");
  } else {
    hipLaunchKernelGGL(nmx_kern_timeosc_w510<8>, dim3(n_items), dim3(64), (size_t)NMX_TO510_LDS_FLOATS * 4, s, *A);
    nmxi_note_kernel("nmx_kern_timeosc_w510<8>");
  }
  return 1;
}

// register-resident scan (Hjorth / Raw / LineLength only): four waves per workgroup, no LDS
__global__ void __launch_bounds__(256) nmx_kern_scan(const NmxTimeOscArgs A, int n_items, int n_windows, int order) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  if (order) nmx_scan_item(A, item % n_windows, item / n_windows);
  else nmx_scan_item(A, item / A.n_channels, item % A.n_channels);
}

extern "C" void nmx_wave_launch_scan(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  static int order = -1;
  if (order < 0) { const char* v = getenv("NMX_SCAN_ORDER"); order = (v && v[0] == '1') ? 1 : 0; }
  hipLaunchKernelGGL(nmx_kern_scan, dim3((n_items + 3) / 4), dim3(256), 0, s, *A, n_items, n_items / A->n_channels, order);
  nmxi_note_kernel("nmx_kern_scan");
}

// dense-first launch: compact LDS layout (more waves per CU); overflowing items are flagged
__global__ void __launch_bounds__(64) nmx_kern_sharp_dense(const NmxSharpArgs A, int n_items) {
  const int item = blockIdx.x;
  const int fi = item % A.n_filters, r = item / A.n_filters;
  nmx_sharp_item_dense(A, r / A.n_channels, r % A.n_channels, fi, (long long)item, nmx_smem_wave);
}

extern "C" void nmx_wave_launch_sharp_dense(const NmxSharpArgs* A, int n_items, hipStream_t s) {
  hipLaunchKernelGGL(nmx_kern_sharp_dense, dim3(n_items), dim3(64), (size_t)A->dz_lds_floats * 4, s, *A, n_items);
  nmxi_note_kernel("nmx_kern_sharp_dense");
}

// items the fused bank kernel could not finish (more than 128 extrema of a kind): generic list code.
// Persistent waves walk the flag array; on the bench workload no flag is ever set.
__global__ void __launch_bounds__(64) nmx_kern_sharp_todo(const NmxSharpArgs A, int n_items, const unsigned char* todo) {
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    if (!todo[item]) continue;
    const int fi = item % A.n_filters, r = item / A.n_filters;
    nmx_sharp_item(A, r / A.n_channels, r % A.n_channels, fi, nmx_smem_wave);
    NMX_SYNC();
  }
}

extern "C" void nmx_wave_launch_sharp_todo(const NmxSharpArgs* A, int n_items, size_t lds, const unsigned char* todo,
                                           hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)nmx_kern_sharp_todo, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int grid = n_items < 256 * 14 ? n_items : 256 * 14;
  hipLaunchKernelGGL(nmx_kern_sharp_todo, dim3(grid), dim3(64), lds, s, *A, n_items, todo);
  nmxi_note_kernel("nmx_kern_sharp_todo");
}

static int waves_per_wg(size_t lds_one) {
  static int k = 0;
  if (!k) {
    const char* v = getenv("NMX_WAVES_PER_WG");
    k = (v && atoi(v) >= 1 && atoi(v) <= 4) ? atoi(v) : 1;
  }
  int kk = k;
  while (kk > 1 && lds_one * kk > 64 * 1024) --kk;
  return kk;
}

extern "C" void nmx_wave_launch_burst_stat(const NmxBurstStatArgs* A, int n_items, size_t lds, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)nmx_kern_burst_stat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int k = waves_per_wg(lds);
  const int slice = (int)((lds / 4 + 3) & ~(size_t)3);
  hipLaunchKernelGGL(nmx_kern_burst_stat, dim3((n_items + k - 1) / k), dim3(64 * k), (size_t)slice * 4 * k, s, *A,
                     n_items, slice);
  nmxi_note_kernel("nmx_kern_burst_stat");
}

extern "C" void nmx_wave_launch_sharp(const NmxSharpArgs* A, int n_items, size_t lds, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)nmx_kern_sharp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int k = waves_per_wg(lds);
  const int slice = (int)((lds / 4 + 3) & ~(size_t)3);
  hipLaunchKernelGGL(nmx_kern_sharp, dim3((n_items + k - 1) / k), dim3(64 * k), (size_t)slice * 4 * k, s, *A,
                     n_items, slice);
  nmxi_note_kernel("nmx_kern_sharp");
}

// nmx_w64.hip -- translation unit of the one-wave FIR kernels (nmx_k_bank_w64*.h): notch, band-pass bank
// (M = 1024 / 1536 channel pairs, M = 2048, M = 4096).  Built ONCE by __graft_entry__.build_lib() with
// -fno-slp-vectorize -DNMX_LDS_ASM=1 -DNMX_W64_NAME=rd64: complex arithmetic is packed explicitly (inline asm with
// operand modifiers, unpaired ds_read_b64); clang's SLP vectoriser on top of that spills.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "nmx_k_bank_w64.h"
#include "nmx_k_bank_w64p.h"
#include "nmx_k_bank_w64x2.h"
#include "nmx_k_bank_w64c.h"
#include "nmx_k_bank_w64d.h"
#include "nmx_k_bank_w64e.h"

#if !defined(NMX_W64_NAME) || !defined(NMX_LDS_ASM)
#error "compile with -DNMX_W64_NAME=rd64 -DNMX_LDS_ASM=1"
#endif
// waves per persistent workgroup (= per CU): 8 leaves the 256-VGPR budget (2 waves/SIMD) that the
// packed-complex formulation needs to stay out of scratch
#ifndef NMX_W64P_WAVES
#define NMX_W64P_WAVES 8
#endif
#define NMX_CAT2(a, b) a##b
#define NMX_CAT(a, b) NMX_CAT2(a, b)
#define NMX_STR2(a) #a
#define NMX_STR(a) NMX_STR2(a)
#define NMX_KNAME(stem, tail) nmxi_note_kernel(stem NMX_STR(NMX_W64_NAME) tail)

extern __shared__ __attribute__((aligned(16))) float nmx_smem_w64[];

// register budgets: the FIR-bank instantiation fits 168 VGPRs (3 waves/SIMD, 12 per CU, 5 spilled
// dwords); the notch instantiation (odd-reflection staging) needs the 256-VGPR budget.
__global__ void __launch_bounds__(64, 3) NMX_CAT(nmx_kern_bank_w64_, NMX_W64_NAME)(const NmxBankW64Args A) {
  const int item = blockIdx.x;
  nmx_bank_w64_item<0, 0, 1>(A, item / A.b.n_channels, item % A.b.n_channels, nmx_smem_w64, nullptr);
}
__global__ void __launch_bounds__(64, 2) NMX_CAT(nmx_kern_notch_w64_, NMX_W64_NAME)(const NmxBankW64Args A) {
  const int item = blockIdx.x;
  nmx_bank_w64_item<1, 0, 0>(A, item / A.b.n_channels, item % A.b.n_channels, nmx_smem_w64, nullptr);
}

// Persistent variant: one workgroup of `nw` waves per CU; the A/B tables of all filters are
// staged in LDS once per workgroup (instead of being re-fetched from L2 for every item: 27 % of
// the kernel's time), then every wave walks its own items with wave-local fences only.
// HIL = 1: Hilbert envelopes of the burst bands inside the kernel (tables after the twiddles in LDS).
// HALF = 1: windows of at most 1024 samples -- the upper half of each inverse transform's outputs is never formed.
// Pipelined persistent kernel (nmx_k_bank_w64p.h): the A / B tables are staged INTERLEAVED ((A_k, B_k) pairs: one
// 8-byte read per point), everything else as below.
template <int HALF>
__global__ void __launch_bounds__(64 * 8) NMX_CAT(nmx_kern_bank_w64pp_, NMX_W64_NAME)(const NmxBankW64Args A0, int n_items,
                                                                                    int x_floats) {
  // (kernel-argument pointer laundered once per item: the plan is re-read with s_load, not hoisted into scalar
  // registers that spill to lanes of a VGPR)
  typedef const NmxBankW64Args __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  float* tab = nmx_smem_w64;
  const int n = NMX_W64_N, tab_floats = ((const NmxBankW64Args*)Ap)->b.n_filters * 2 * n;
  {
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    for (int i = threadIdx.x; i < tab_floats; i += blockDim.x) {
      const int fi = i / (2 * n), j = i - fi * 2 * n;
      tab[i] = (j & 1) ? A.Hd[fi][j >> 1] : A.Hs[fi][j >> 1];
    }
    for (int i = threadIdx.x; i < NMX_W64_TWL_FLOATS; i += blockDim.x) tab[tab_floats + i] = A.twl[i];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  float* mine = nmx_smem_w64 + tab_floats + NMX_W64_TWL_FLOATS + wave * x_floats;
#pragma nounroll
  for (int item = blockIdx.x * nw + wave; item < n_items; item += gridDim.x * nw) {
    asm volatile("" : "+s"(Ap));
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    nmx_bank_w64_item_pipe<HALF>(A, item / A.b.n_channels, item % A.b.n_channels, mine, tab);
  }
}

// Notch, four items per workgroup: the filter's A / B tables and the twiddles are staged in LDS once per
// FOUR items (the one-wave-per-workgroup kernel fetches ~27 KB of tables from L2 per item); no item loop,
// so none of the scalar-register pressure of the persistent form.
__global__ void __launch_bounds__(256, 3) NMX_CAT(nmx_kern_notch_w64q_, NMX_W64_NAME)(const NmxBankW64Args A, int n_items,
                                                                                      int x_floats) {
  float* tab = nmx_smem_w64;
  const int n = NMX_W64_N, tab_floats = 2 * n;
  for (int i = threadIdx.x; i < tab_floats; i += 256) tab[i] = i < n ? A.Hs[0][i] : A.Hd[0][i - n];
  for (int i = threadIdx.x; i < NMX_W64_TWL_FLOATS; i += 256) tab[tab_floats + i] = A.twl[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  nmx_bank_w64_item<1, 1, 0>(A, item / A.b.n_channels, item % A.b.n_channels,
                             nmx_smem_w64 + tab_floats + NMX_W64_TWL_FLOATS + wave * x_floats, tab);
}

// waves of the persistent notch workgroup: ONE workgroup of twelve waves per CU (3 waves / SIMD at 168 VGPRs) shares one
// copy of the 20 KB of tables -- measured on one lease with the reflection table: 12 waves 0.96 ms, 6 (two workgroups per
// CU) 1.15 - 1.20, 4 (only two workgroups fit: 8 waves) 1.22
#ifndef NMX_NOTCH_QP_WAVES
#define NMX_NOTCH_QP_WAVES 12
#endif
// Persistent form of the above: one workgroup per CU (filter tables, twiddles and the reflection table: 20 KB, one
// exchange tile per wave) walks the items -- the tables are staged once
// per workgroup instead of once per four items (the staging + its barrier + the workgroup launch were a quarter of
// the four-item kernel's time).  The kernel-argument pointer is laundered once per iteration so that the plan is
// re-read with s_load instead of being hoisted into (spilled) scalar registers (nmx_wave.hip).
__global__ void __launch_bounds__(64 * NMX_NOTCH_QP_WAVES, 3) NMX_CAT(nmx_kern_notch_w64qp_, NMX_W64_NAME)(const NmxBankW64Args A0, int n_items,
                                                                                       int x_floats) {
  typedef const NmxBankW64Args __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  float* tab = nmx_smem_w64;
  const int n = NMX_W64_N, tab_floats = 2 * n;
  {
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    for (int i = threadIdx.x; i < tab_floats; i += 64 * NMX_NOTCH_QP_WAVES) tab[i] = i < n ? A.Hs[0][i] : A.Hd[0][i - n];
    for (int i = threadIdx.x; i < NMX_W64_TWL_FLOATS; i += 64 * NMX_NOTCH_QP_WAVES) tab[tab_floats + i] = A.twl[i];
  }
  unsigned* rtab = (unsigned*)(nmx_smem_w64 + tab_floats + NMX_W64_TWL_FLOATS);   // [16][64] (4 KB)
  nmx_w64_reflect_table(((const NmxBankW64Args*)Ap)->b, rtab, (int)threadIdx.x, 64 * NMX_NOTCH_QP_WAVES);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* mine = nmx_smem_w64 + tab_floats + NMX_W64_TWL_FLOATS + 1024 + wave * x_floats;
#pragma nounroll
  for (int item = blockIdx.x * NMX_NOTCH_QP_WAVES + wave; item < n_items; item += (int)gridDim.x * NMX_NOTCH_QP_WAVES) {
    asm volatile("" : "+s"(Ap));
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    nmx_bank_w64_item<1, 1, 0>(A, item / A.b.n_channels, item % A.b.n_channels, mine, tab, rtab);
    NMX_WAVE_FENCE();
  }
}

// M = 4096 (nmx_k_bank_w64x2.h): persistent workgroups of `nw` waves; LDS = tables of the first n_tab filters,
// pass B / C twiddles, w^k, one exchange tile per wave
template <int HALF>
__global__ void __launch_bounds__(64 * 8) NMX_CAT(nmx_kern_bank_w64x2_, NMX_W64_NAME)(const NmxBankW64Args A0, int n_items,
                                                                                    int x_floats, int n_tab) {
  // (the kernel-argument pointer is laundered once per item: the plan -- eight filters' worth of descriptors -- is then
  // re-read with s_load instead of being hoisted into scalar registers that spill to v_writelane / v_readlane, 108 of
  // them in the first form of this loop)
  typedef const NmxBankW64Args __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  float* tab = nmx_smem_w64;
  const int tab_floats = n_tab * 4096;
  {
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    for (int i = threadIdx.x; i < tab_floats; i += blockDim.x) tab[i] = A.Hs[i >> 12][i & 4095];
    for (int i = threadIdx.x; i < NMX_W64_TWL_FLOATS; i += blockDim.x) tab[tab_floats + i] = A.twl[i];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) tab[tab_floats + NMX_W64_TWL_FLOATS + i] = A.tw2[i];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  float* mine = nmx_smem_w64 + tab_floats + NMX_W64_TWL_FLOATS + 2048 + wave * x_floats;
#pragma nounroll
  for (int item = blockIdx.x * nw + wave; item < n_items; item += gridDim.x * nw) {
    asm volatile("" : "+s"(Ap));
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    nmx_bank_w64x2_item<HALF>(A, item / A.b.n_channels, item % A.b.n_channels, mine, tab, n_tab);
  }
}

extern "C" int NMX_CAT(nmx_w64x2_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, int n_cu, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64x2_, NMX_W64_NAME)<0>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64x2_, NMX_W64_NAME)<1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  static int nw_env = 0;
  if (!nw_env) nw_env = 8;
  const int nw = nw_env, x_floats = A->lds_floats;
  int n_tab = (160 * 1024 / 4 - NMX_W64_TWL_FLOATS - 2048 - nw * x_floats) / 4096;
  if (n_tab > A->b.n_filters) n_tab = A->b.n_filters;
  if (n_tab < 0) return 0;
  const size_t lds = (size_t)(n_tab * 4096 + NMX_W64_TWL_FLOATS + 2048 + nw * x_floats) * 4;
  int grid = n_cu > 0 ? n_cu : 256;
  if (grid * nw > n_items) grid = (n_items + nw - 1) / nw;
  if (A->b.W <= 2048) {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64x2_, NMX_W64_NAME)<1>), dim3(grid), dim3(64 * nw), lds, s, *A, n_items, x_floats, n_tab);
    NMX_KNAME("nmx_kern_bank_w64x2_", "<1>");
  } else {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64x2_, NMX_W64_NAME)<0>), dim3(grid), dim3(64 * nw), lds, s, *A, n_items, x_floats, n_tab);
    NMX_KNAME("nmx_kern_bank_w64x2_", "<0>");
  }
  return 1;
}

// M = 1536, one wave per (window, channel pair) (nmx_k_bank_w64c.h): workgroups of `nw` waves; LDS = the real spectra of
// all filters, the pass-A twiddles, one exchange tile per wave.  A wave walks a CONTIGUOUS run of `chunk` items in the
// order (channel pair, window): consecutive items are consecutive hops of the same two channels, whose windows
// overlap by W - hop samples -- after the first item of a run most of the window comes from L2.
__global__ void __launch_bounds__(64 * 8) NMX_CAT(nmx_kern_bank_w64c_, NMX_W64_NAME)(const NmxBankW64Args A, int n_windows,
                                                                                   int n_pairs, int chunk) {
  float* tab = nmx_smem_w64;
  const int hf = A.b.n_filters * NMX_W64C_H_FLOATS;
  for (int i = threadIdx.x; i < hf; i += blockDim.x) tab[i] = A.hc[i];
  for (int i = threadIdx.x; i < NMX_W64C_TWA_FLOATS; i += blockDim.x) tab[hf + i] = A.twc[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  NmxW64cLane Ln;
  nmx_w64c_lane_setup(Ln, tab + hf + NMX_W64C_TWA_FLOATS + wave * NMX_W64C_TILE_FLOATS, tab + hf,
                      A.twc + NMX_W64C_TWA_FLOATS, (int)(threadIdx.x & 63));
  const int q0 = (blockIdx.x * nw + wave) * chunk;
  const int q1 = q0 + chunk < n_pairs ? q0 + chunk : n_pairs;
#pragma nounroll
  for (int q = q0; q < q1; ++q) {
    const int cp = q / n_windows;
    nmx_bank_w64c_item(A, q - cp * n_windows, 2 * cp, Ln, tab);
  }
}

// M = 1024, one wave per (window, channel pair) (nmx_k_bank_w64d.h): LDS = real spectra of all filters, pass B / C
// twiddles, one exchange tile per wave; the same contiguous runs of hops per wave
template <int HALF>
__global__ void __launch_bounds__(64 * (HALF ? 12 : 8)) NMX_CAT(nmx_kern_bank_w64d_, NMX_W64_NAME)(const NmxBankW64Args A, int n_windows,
                                                                                   int n_pairs, int chunk, int x_floats) {
  float* tab = nmx_smem_w64;
  const int hf = A.b.n_filters * NMX_W64D_H_FLOATS;
  for (int i = threadIdx.x; i < hf; i += blockDim.x) tab[i] = A.hc[i];
  for (int i = threadIdx.x; i < NMX_W64_TWL_FLOATS; i += blockDim.x) tab[hf + i] = A.twl[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  float* mine = tab + hf + NMX_W64_TWL_FLOATS + wave * x_floats;
  const int q0 = (blockIdx.x * nw + wave) * chunk;
  const int q1 = q0 + chunk < n_pairs ? q0 + chunk : n_pairs;
#pragma nounroll
  for (int q = q0; q < q1; ++q) {
    const int cp = q / n_windows;
    nmx_bank_w64d_item<HALF>(A, q - cp * n_windows, 2 * cp, mine, tab);
  }
}

extern "C" int NMX_CAT(nmx_w64d_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, int n_cu, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64d_, NMX_W64_NAME)<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64d_, NMX_W64_NAME)<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int C = A->b.n_channels, n_windows = n_items / C, n_pairs = n_windows * ((C + 1) / 2);
  const int x_floats = A->lds_floats;
  const int fixed = A->b.n_filters * NMX_W64D_H_FLOATS + NMX_W64_TWL_FLOATS;
  int nw = (160 * 1024 / 4 - fixed) / x_floats;
  if (nw < 6) return 0;
  static int want = 0;
  if (!want) want = 12;
  const int cap = A->b.W <= 512 ? want : (want < 8 ? want : 8);   // W <= 512: 114 VGPRs, three waves per SIMD fit
  if (nw > cap) nw = cap;
  if (n_pairs < 2048) nw = 2;
  const size_t lds = (size_t)(fixed + nw * x_floats) * 4;
  int grid = n_cu > 0 ? n_cu : 256;
  if (grid * nw > n_pairs) grid = (n_pairs + nw - 1) / nw;
  const int chunk = (n_pairs + grid * nw - 1) / (grid * nw);
  if (A->b.W <= 512) {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64d_, NMX_W64_NAME)<1>), dim3(grid), dim3(64 * nw), lds, s, *A, n_windows, n_pairs, chunk, x_floats);
    NMX_KNAME("nmx_kern_bank_w64d_", "<1>");
  } else {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64d_, NMX_W64_NAME)<0>), dim3(grid), dim3(64 * nw), lds, s, *A, n_windows, n_pairs, chunk, x_floats);
    NMX_KNAME("nmx_kern_bank_w64d_", "<0>");
  }
  return 1;
}

// M = 2048, one wave per (window, channel pair) (nmx_k_bank_w64e.h): PAD = 0 the "same" FIR bank of the filters the
// M = 1536 kernel cannot take, PAD = 1 the notch (odd-reflected window, one filter, the window back to HBM).  LDS = the
// real spectra of the filters, the pass-A twiddles, one 18 KiB exchange tile per wave; contiguous runs of hops per wave.
template <int PAD, int WC = 0, int HC = 0>
__global__ void __launch_bounds__(64 * 8) NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)(const NmxBankW64Args A0, int n_windows,
                                                                                   int n_pairs, int chunk) {
  // (kernel-argument pointer laundered once per item: the plan is re-read with s_load, not hoisted into scalar
  // registers that spill to lanes of a VGPR)
  typedef const NmxBankW64Args __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  float* tab = nmx_smem_w64;
  const int hf = ((const NmxBankW64Args*)Ap)->b.n_filters * NMX_W64E_H_FLOATS;
  NmxW64cLane Ln;
  unsigned rt[16];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  {
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    for (int i = threadIdx.x; i < hf; i += blockDim.x) tab[i] = A.hc[i];
    for (int i = threadIdx.x; i < NMX_W64E_TWA_FLOATS; i += blockDim.x) tab[hf + i] = A.twc[i];
    nmx_w64c_lane_setup(Ln, tab + hf + NMX_W64E_TWA_FLOATS + wave * NMX_W64E_TILE_FLOATS, tab + hf,
                        A.twc + NMX_W64E_TWA_FLOATS, (int)(threadIdx.x & 63));
    if (PAD && !WC) nmx_w64e_reflect_lane(A.b, (int)(threadIdx.x & 63), rt);
  }
  __syncthreads();
  const int q0 = (blockIdx.x * nw + wave) * chunk;
  const int q1 = q0 + chunk < n_pairs ? q0 + chunk : n_pairs;
#pragma nounroll
  for (int q = q0; q < q1; ++q) {
    asm volatile("" : "+s"(Ap));
    const NmxBankW64Args& A = *(const NmxBankW64Args*)Ap;
    const int cp = q / n_windows;
    nmx_bank_w64e_item<PAD, WC, HC>(A, q - cp * n_windows, 2 * cp, Ln, tab, rt);
  }
}

// Would the dispatcher of nmx_api.hip (be_launch_bank_w64) end up in a kernel that adds the carried offset on load
// (NmxBankArgs::dcf: the channel-pair kernels of nmx_k_bank_w64c.h / nmx_k_bank_w64e.h)?  The same conditions as the
// launchers below; anything else reads a copy of the windows with the offset added back (nmx_engine_run.inc).
extern "C" int NMX_CAT(nmx_w64_takes_dc_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items) {
  (void)n_items;
  if (A->tw2 || !A->hc || A->b.pad_mode != 0 || A->b.n_filters < 1) return 0;
  if (A->pair_m == 2048) {
    if (A->b.W > 1024 || (A->b.bp_features & 6u) || !A->twc) return 0;
    return (160 * 1024 / 4 - (A->b.n_filters * NMX_W64E_H_FLOATS + NMX_W64E_TWA_FLOATS)) / NMX_W64E_TILE_FLOATS >= 4;
  }
  if (A->pair_m == 1536)
    return (160 * 1024 / 4 - (A->b.n_filters * NMX_W64C_H_FLOATS + NMX_W64C_TWA_FLOATS)) / NMX_W64C_TILE_FLOATS >= 6;
  return 0;
}

// returns 0 when the configuration does not fit (caller falls back to the one-channel M = 2048 kernels)
extern "C" int NMX_CAT(nmx_w64e_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, int n_cu, hipStream_t s) {
  const bool pad = A->b.pad_mode != 0;
  if (A->b.W > 1024 || (A->b.bp_features & 6u) || !A->hc || !A->twc || A->b.n_filters < 1) return 0;
  if (pad && (A->b.n_filters != 1 || A->b.W + 2 * A->b.pad_half > NMX_W64E_M)) return 0;
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)<1, 1000, 499>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int C = A->b.n_channels, n_windows = n_items / C, n_pairs = n_windows * ((C + 1) / 2);
  const int fixed = A->b.n_filters * NMX_W64E_H_FLOATS + NMX_W64E_TWA_FLOATS;
  int nw = (160 * 1024 / 4 - fixed) / NMX_W64E_TILE_FLOATS;
  if (nw < 4) return 0;
  static int want = 0;
  if (!want) want = 8;
  if (nw > want) nw = want;
  if (n_pairs < 2048) nw = 2;   // a hop or two: spread the few items over many CUs
  const size_t lds = (size_t)(fixed + nw * NMX_W64E_TILE_FLOATS) * 4;
  int grid = n_cu > 0 ? n_cu : 256;
  if (grid * nw > n_pairs) grid = (n_pairs + nw - 1) / nw;
  const int chunk = (n_pairs + grid * nw - 1) / (grid * nw);
  if (pad && A->b.W == 1000 && A->b.pad_half == 499 && A->b.n_edge >= 499) {   // the default notch: 1 kHz x 1 s windows, 999 taps
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)<1, 1000, 499>), dim3(grid), dim3(64 * nw), lds, s, *A, n_windows, n_pairs, chunk);
    NMX_KNAME("nmx_kern_bank_w64e_", "<1, 1000, 499>");
  } else if (pad) {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)<1>), dim3(grid), dim3(64 * nw), lds, s, *A, n_windows, n_pairs, chunk);
    NMX_KNAME("nmx_kern_bank_w64e_", "<1>");
  } else {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64e_, NMX_W64_NAME)<0>), dim3(grid), dim3(64 * nw), lds, s, *A, n_windows, n_pairs, chunk);
    NMX_KNAME("nmx_kern_bank_w64e_", "<0>");
  }
  return 1;
}

// returns 0 when the tables do not fit next to at least six tiles (caller falls back to the M = 2048 kernels)
extern "C" int NMX_CAT(nmx_w64c_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, int n_cu, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen))
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64c_, NMX_W64_NAME),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int C = A->b.n_channels, n_windows = n_items / C, n_pairs = n_windows * ((C + 1) / 2);
  const int fixed = A->b.n_filters * NMX_W64C_H_FLOATS + NMX_W64C_TWA_FLOATS;
  int nw = (160 * 1024 / 4 - fixed) / NMX_W64C_TILE_FLOATS;
  if (nw < 6) return 0;
  if (nw > 8) nw = 8;
  if (n_pairs < 2048) nw = 2;   // a hop or two: spread the few items over many CUs
  const size_t lds = (size_t)(fixed + nw * NMX_W64C_TILE_FLOATS) * 4;
  int grid = n_cu > 0 ? n_cu : 256;
  if (grid * nw > n_pairs) grid = (n_pairs + nw - 1) / nw;
  const int chunk = (n_pairs + grid * nw - 1) / (grid * nw);
#ifdef NMX_DEBUG_NO_YB
  static int launches = 0;
  NmxBankW64Args B = *A;
  if (++launches > 6) B.yb_out = nullptr;
  A = &B;
#endif
  hipLaunchKernelGGL(NMX_CAT(nmx_kern_bank_w64c_, NMX_W64_NAME), dim3(grid), dim3(64 * nw), lds, s, *A, n_windows, n_pairs, chunk);
  NMX_KNAME("nmx_kern_bank_w64c_", "");
  return 1;
}

// returns 0 when the configuration does not fit (caller falls back to one wave per workgroup)
extern "C" int NMX_CAT(nmx_w64q_launch_notch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, hipStream_t s) {
  if (A->b.pad_mode == 0 || A->b.n_filters != 1 || !A->twl) return 0;
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_notch_w64q_, NMX_W64_NAME),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int x_floats = A->lds_floats;
  const size_t lds = (size_t)(2 * NMX_W64_N + NMX_W64_TWL_FLOATS + 4 * x_floats) * 4;
  if (n_items >= 3 * 256 * 4 * 8) {   // eight items or more per wave: persistent workgroups
    static unsigned long long seen_p = 0;
    if (nmx_first_on_device(seen_p))
      (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_notch_w64qp_, NMX_W64_NAME),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t ldsp = (size_t)(2 * NMX_W64_N + NMX_W64_TWL_FLOATS + 1024 + NMX_NOTCH_QP_WAVES * x_floats) * 4;
    hipLaunchKernelGGL(NMX_CAT(nmx_kern_notch_w64qp_, NMX_W64_NAME), dim3((12 / NMX_NOTCH_QP_WAVES) * 256), dim3(64 * NMX_NOTCH_QP_WAVES), ldsp, s, *A, n_items, x_floats);
    NMX_KNAME("nmx_kern_notch_w64qp_", "");
    return 1;
  }
  hipLaunchKernelGGL(NMX_CAT(nmx_kern_notch_w64q_, NMX_W64_NAME), dim3((n_items + 3) / 4), dim3(256), lds, s, *A, n_items,
                     x_floats);
  NMX_KNAME("nmx_kern_notch_w64q_", "");
  return 1;
}

// M = 2048 band-pass bank, persistent 8-wave workgroups with the software-pipelined item (nmx_k_bank_w64p.h).
// Returns 0 when the configuration does not fit (caller falls back to one wave per workgroup).
extern "C" int NMX_CAT(nmx_w64p_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, int n_cu, hipStream_t s) {
  if ((A->b.bp_features & 6u) || A->b.pad_mode != 0 || !A->twl || (A->b.W & 1)) return 0;
  const int x_floats = A->lds_floats;                  // per-wave exchange tile (+ scratch)
  const int tab_floats = A->b.n_filters * 2 * NMX_W64_N;
  const int nw = 8;
  if ((160 * 1024 / 4 - tab_floats - NMX_W64_TWL_FLOATS) / x_floats < nw) return 0;
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64pp_, NMX_W64_NAME)<0>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64pp_, NMX_W64_NAME)<1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const size_t lds = (size_t)(tab_floats + NMX_W64_TWL_FLOATS + nw * x_floats) * 4;
  int grid = n_cu > 0 ? n_cu : 256;
  if (grid * nw > n_items) grid = (n_items + nw - 1) / nw;
  if (A->b.W <= 1024) {   // the upper half of every inverse transform's outputs is never formed
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64pp_, NMX_W64_NAME)<1>), dim3(grid), dim3(64 * nw), lds, s, *A, n_items, x_floats);
    NMX_KNAME("nmx_kern_bank_w64pp_", "<1>");
  } else {
    hipLaunchKernelGGL((NMX_CAT(nmx_kern_bank_w64pp_, NMX_W64_NAME)<0>), dim3(grid), dim3(64 * nw), lds, s, *A, n_items, x_floats);
    NMX_KNAME("nmx_kern_bank_w64pp_", "<0>");
  }
  return 1;
}

extern "C" void NMX_CAT(nmx_w64_launch_, NMX_W64_NAME)(const NmxBankW64Args* A, int n_items, size_t lds,
                                                       hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_bank_w64_, NMX_W64_NAME),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)NMX_CAT(nmx_kern_notch_w64_, NMX_W64_NAME),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (A->b.pad_mode == 0) {
    hipLaunchKernelGGL(NMX_CAT(nmx_kern_bank_w64_, NMX_W64_NAME), dim3(n_items), dim3(64), lds, s, *A);
    NMX_KNAME("nmx_kern_bank_w64_", "");
  } else {
    hipLaunchKernelGGL(NMX_CAT(nmx_kern_notch_w64_, NMX_W64_NAME), dim3(n_items), dim3(64), lds, s, *A);
    NMX_KNAME("nmx_kern_notch_w64_", "");
  }
}

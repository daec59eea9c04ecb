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
#include "nmx_k_burst_stat_reg.h"
#include "nmx_k_scan.h"
#include "nmx_k_td.h"
#include "nmx_k_timeosc_w1000.h"
#include "nmx_k_timeosc_w510.h"
#include "nmx_k_sharpwave.h"

extern __shared__ __attribute__((aligned(16))) float nmx_smem_wave[];

// One-wave-per-item kernels can be launched with several waves per workgroup (each wave its own item and LDS slice, no
// barrier between them).  Measured on the MI355X (profiles/r04_waves_per_wg.txt): the sharp-wave kernels -- 30+ KB of
// code, waves at unrelated program counters -- gain 13 % (1000-sample windows) to 24 % (BASELINE config 5) with TWO
// waves per workgroup, which start together and fetch the same instructions; four are no better; the transform kernels
// (time / oscillatory, Hilbert) do not gain, and lose when they overlap the bursts chain.  `dflt` is that choice,
// NMX_WAVES_PER_WG = 1 .. 4 overrides it for every kernel.
static int waves_per_wg(size_t lds_one, int dflt = 1) {
  const char* v = getenv("NMX_WAVES_PER_WG");   // (read per launch: the GPU tests switch it inside one process)
  const int k = (v && atoi(v) >= 1 && atoi(v) <= 4) ? atoi(v) : 0;
  int kk = k ? k : dflt;
  while (kk > 1 && lds_one * kk > 64 * 1024) --kk;
  return kk;
}
// (item, LDS slice) of this wave in a launch of `blockDim.x / 64` waves per workgroup; slice in floats
#define NMX_WAVE_ITEM(item, smem, n_items, slice)                                                   \
  const int wave_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                        \
  const int item = (int)blockIdx.x * (int)(blockDim.x >> 6) + wave_;                                \
  if (item >= (n_items)) return;                                                                    \
  float* smem = nmx_smem_wave + wave_ * (slice)

__global__ void __launch_bounds__(256) nmx_kern_burst_stat(const NmxBurstStatArgs A, int n_items, int slice) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPR
  const int item = blockIdx.x * (blockDim.x >> 6) + wave;
  if (item >= n_items) return;
  const int bi = item % A.n_bands, r = item / A.n_bands;
  nmx_burst_stat_item(A, r / A.n_channels, r % A.n_channels, bi, nmx_smem_wave + wave * slice);
}

// the envelope of one item in the registers of its wave (nmx_k_burst_stat_reg.h): W % 4 == 0, W <= 64 CH; no LDS
template <int CH>
__global__ void __launch_bounds__(256) nmx_kern_burst_stat_reg(const NmxBurstStatArgs A, int n_items) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  const int bi = item % A.n_bands, r = item / A.n_bands;
  nmx_burst_stat_item_reg<CH>(A, r / A.n_channels, r % A.n_channels, bi);
}

__global__ void __launch_bounds__(256) nmx_kern_sharp(const NmxSharpArgs A, int n_items, int slice) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPR
  const int item = blockIdx.x * (blockDim.x >> 6) + wave;
  if (item >= n_items) return;
  const int fi = item % A.n_filters, r = item / A.n_filters;
  nmx_sharp_item(A, r / A.n_channels, r % A.n_channels, fi, nmx_smem_wave + wave * slice);
}

// threshold walk, steady regime: one wave per (channel, band); NR = registers per lane for a hop's new samples, LL = the
// top-K list in LDS for the launch (nmx_k_bursts.h)
template <int NR, bool LL>
__global__ void __launch_bounds__(64) nmx_kern_burst_thr_wave(const NmxBurstThrArgs A) {
  __builtin_amdgcn_s_setprio(3);   // sequential and on the critical path: win issue arbitration
  const int item = blockIdx.x;
  nmx_burst_thr_wave_item<NR, LL>(A, item / A.n_bands, item % A.n_bands, nmx_smem_wave);
}

extern "C" void nmx_wave_launch_burst_thr(const NmxBurstThrArgs* A, int n_items, hipStream_t s, long long windows_seen) {
  const char* v_ll = getenv("NMX_THR_LIST_LDS");   // (read per launch: the tests run both forms in one process)
  const bool lds_list = !(v_ll && v_ll[0] == '0');
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)nmx_kern_burst_thr_wave<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)nmx_kern_burst_thr_wave<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int nr = A->overlap <= 128 ? 2 : 4;
  const size_t base = (size_t)((nr == 2 ? NMX_THRW_LDS_FLOATS_NR(2) : NMX_THRW_LDS_FLOATS_NR(4)) + A->K / 64 + 4) * 4;
  const size_t with_list = (size_t)((nr == 2 ? NMX_THRW_LDS_FLOATS_OF(2, true) : NMX_THRW_LDS_FLOATS_OF(4, true)) + A->K / 64 + 4) * 4 +
                           (size_t)A->K * 4;
  // The list in LDS pays while the stream is YOUNG: the ring has just filled, a quarter of every hop's samples still enters
  // the list and a flush is due every ~15 hops (a fresh 120 s stream: 25.4 -> 22.2 ms end to end).  After thousands of hops
  // the kept minimum has risen, flushes are rare, and 56 KB of LDS per walk only take occupancy from the throughput
  // kernels running next to it (the bench's steady state: 6.59 -> 6.87 ms per step) -- then the list stays in L2.
  // ... only while the walks still fit the chip in two rounds (the default history: 52 KB, three walks per CU, 1536 series
  // in two rounds), and only for hops of <= 128 samples: config 3 (2 kHz, 200 samples per hop, the four-register walk with
  // its 512-entry pending list) measured 1.46 -> 2.2 ms per 256 hops with its list in LDS
  const long long per_round = 256LL * (long long)((160 * 1024) / with_list);
  const bool ll = lds_list && nr == 2 && with_list <= 80 * 1024 && windows_seen < 4096 && (long long)n_items <= 2 * per_round;
  const size_t lds = ll ? with_list : base;
  if (nr == 2) {
    if (ll) hipLaunchKernelGGL((nmx_kern_burst_thr_wave<2, true>), dim3(n_items), dim3(64), lds, s, *A);
    else hipLaunchKernelGGL((nmx_kern_burst_thr_wave<2, false>), dim3(n_items), dim3(64), lds, s, *A);
    nmxi_note_kernel(ll ? "nmx_kern_burst_thr_wave<2, true>" : "nmx_kern_burst_thr_wave<2, false>");
  } else {
    hipLaunchKernelGGL((nmx_kern_burst_thr_wave<4, false>), dim3(n_items), dim3(64), lds, s, *A);
    nmxi_note_kernel("nmx_kern_burst_thr_wave<4, false>");
  }
}

// Hilbert envelope of length-1000 series, one wave per series (wave-level 500-point transforms)
__global__ void __launch_bounds__(256) nmx_kern_hilbert_w500(const NmxHilbertArgs A, int n_items) {
  NMX_WAVE_ITEM(item, smem, n_items, NMX_W500_LDS_FLOATS);
  nmx_hilbert_w500_item(A, (long long)item, smem);
}

extern "C" void nmx_wave_launch_hilbert_w500(const NmxHilbertArgs* A, long long n_items, hipStream_t s) {
  const int k = waves_per_wg((size_t)NMX_W500_LDS_FLOATS * 4);
  hipLaunchKernelGGL(nmx_kern_hilbert_w500, dim3((unsigned)((n_items + k - 1) / k)), dim3(64 * k),
                     (size_t)NMX_W500_LDS_FLOATS * 4 * k, s, *A, (int)n_items);
  nmxi_note_kernel("nmx_kern_hilbert_w500");
}

// Hilbert envelope of length-2000 series (BASELINE config 3), one wave per series, four per workgroup (16 KB of LDS each)
__global__ void __launch_bounds__(256) nmx_kern_hilbert_w1000(const NmxHilbertArgs A, long long n_items) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long item = (long long)blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  nmx_hilbert_w1000_item(A, item, nmx_smem_wave + wave * NMX_W1000_LDS_FLOATS);
}

extern "C" void nmx_wave_launch_hilbert_w1000(const NmxHilbertArgs* A, long long n_items, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen))
    (void)hipFuncSetAttribute((const void*)nmx_kern_hilbert_w1000, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(nmx_kern_hilbert_w1000, dim3((unsigned)((n_items + 3) / 4)), dim3(256), (size_t)4 * NMX_W1000_LDS_FLOATS * 4, s, *A, n_items);
  nmxi_note_kernel("nmx_kern_hilbert_w1000");
}

// time-domain + FFT / Welch / STFT band means of the default shape, one wave per (window, channel)
template <int NB, unsigned SPEC = 0>
__global__ void __launch_bounds__(256) nmx_kern_timeosc_w1000(const NmxTimeOscArgs A, int n_items, int slice) {
  NMX_WAVE_ITEM(item, smem, n_items, slice);
  const int w = nmx_uniform_i(item / A.n_channels), c = nmx_uniform_i(item % A.n_channels);
  NmxW500TwReg T;
  T.load(A.w500_tab, (int)(threadIdx.x & 63));
  NmxTdRegs R;
  nmx_td_load<1000>(A, w, c, R);
  nmx_timeosc_w1000_body<NB, false, NmxW500TwReg, false, SPEC>(A, w, c, R, T, smem);
}

// The same without an STFT and with low bands only (nmx_timeosc_w1000_low_ok: the default 4 - 35 Hz bands, BASELINE
// config[1]): PERSISTENT waves (one-wave workgroups, grid = what the chip holds at once) walk the items with stride
// gridDim, and the 16-byte loads of a wave's NEXT window are issued before it works on the current one -- the HBM
// latency of a window (4 KB per wave, distinct per item in SURVEY 8d's Mode A) hides behind the arithmetic of the item
// instead of idling the wave (one-item-per-workgroup form: SQ_WAIT_ANY 45 % of the wave cycles).  What it takes:
//   * vmcnt retires IN ORDER, so the item must not wait for ANY vector-memory load issued after the prefetch: the small
//     loads of the current item (successors, row ends) go out before it, the real-transform twiddles of the bins come
//     from a 102-entry LDS copy, nothing is spilled (3 waves per SIMD: 168 VGPRs), and the prefetch is UNCONDITIONAL
//     (a load issued on some paths only forces s_waitcnt vmcnt(0) at the next use of anything loaded earlier);
//   * the plan (band tables, output columns: ~100 dwords) must NOT be hoisted out of the item loop into scalar
//     registers -- it does not fit, and spilled scalars come back as v_readlane VALU instructions (500 of them in a
//     first build): the kernel-argument pointer is laundered once per iteration, so every use re-reads its dwords with
//     s_load from the constant cache, as the one-item-per-workgroup kernel does.
//   * SPEC: the kernel is also built for two fixed feature sets (nmx_k_timeosc_w1000.h: NMX_TOW_SPEC_*): with the
//     feature tests folded away an item is ~120 scalar instructions and branches shorter, and those share the wave's
//     issue slots with everything else.
template <int NB, unsigned SPEC = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4)))
nmx_kern_timeosc_w1000_low(const NmxTimeOscArgs A0, int n_items) {
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  const NmxTimeOscArgs& A = *(const NmxTimeOscArgs*)Ap;
  const int lane = (int)(threadIdx.x & 63);
  NmxW500TwReg T;
  T.load(A.w500_tab, lane);
  {
    const float2* twr = (A.fft.enabled ? A.fft : A.welch).fft.twr;
    float2* twl = (float2*)(nmx_smem_wave + NMX_TOW_LOW_TWL_OFF);
    twl[lane] = twr[lane];
    if (lane < 38) twl[64 + lane] = twr[64 + lane];
  }
  const int C = A.n_channels, step = (int)gridDim.x;
  int item = (int)blockIdx.x;
  if (item >= n_items) return;
  NmxTdRegs R;
  nmx_f4 Xn[4];
  int w = nmx_uniform_i(item / C), c = nmx_uniform_i(item - w * C);
  nmx_td_load_x<1000>(A, w, c, R);
  NMX_WAVE_FENCE();
#pragma nounroll
  for (;;) {
    const int nxt = item + step;
    int wn = 0, cn = 0;
    nmx_td_rest_from_x<1000>(R);
    {
      const int pf = nxt < n_items ? nxt : item;   // (the last iteration re-reads its own item)
      wn = nmx_uniform_i(pf / C); cn = nmx_uniform_i(pf - wn * C);
      NmxTdRegs Rn;
      nmx_td_load_x<1000>(A, wn, cn, Rn);
      Xn[0] = Rn.x[0]; Xn[1] = Rn.x[1]; Xn[2] = Rn.x[2]; Xn[3] = Rn.x[3];
    }
    asm volatile("" : "+s"(Ap));
    nmx_timeosc_w1000_body<NB, true, NmxW500TwReg, false, SPEC>(*(const NmxTimeOscArgs*)Ap, w, c, R, T, nmx_smem_wave);
    NMX_WAVE_FENCE();
    if (nxt >= n_items) break;
    item = nxt; w = wn; c = cn;
    R.x[0] = Xn[0]; R.x[1] = Xn[1]; R.x[2] = Xn[2]; R.x[3] = Xn[3];
  }
}


// does the kernel be_launch_timeosc (nmx_api.hip) would pick take the carried offset (NmxTimeOscArgs::dcf)?  Every one but
// the two special-shape kernels below (they read a copy of the windows with the offset added back: nmx_engine_run.inc)
extern "C" int nmx_wave_timeosc_takes_dc(const NmxTimeOscArgs* A) {
  if (!A->fft.enabled && !A->welch.enabled && !A->stft.enabled) return 1;
  if (A->w500_tab && nmx_timeosc_w1000_ok(*A)) return 1;
  if (A->w500_tab && nmx_timeosc_stft500_ok(*A)) return 0;
  if (A->w510_tab && nmx_timeosc_w510_ok(*A, A->w510_tab)) return 0;
  return 1;
}

// returns 0 when the configuration needs the generic kernel
extern "C" int nmx_wave_launch_timeosc_w1000(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  if (!nmx_timeosc_w1000_ok(*A)) return 0;
  static int n_cu = 0, want = 0, low_ok = 1;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    want = 12;
    const char* u = getenv("NMX_TOW_PERSISTENT");
    low_ok = !(u && u[0] == '0');
  }
  if (low_ok && nmx_timeosc_w1000_low_ok(*A)) {
    // resident waves per CU: 3 per SIMD (168 VGPRs).  The channel of a wave's items stays fixed -- and with it the XCD
    // whose L2 holds the overlapping windows -- when the stride is a multiple of the channel count
    int grid = n_cu * want;
    const int C = A->n_channels;
    if (grid > C && grid % C) grid -= grid % C;
    if (grid > n_items) grid = n_items;
    const size_t lds = (size_t)NMX_TOW_LOW_LDS_FLOATS * 4;
    const unsigned spec = nmx_tow_spec(*A);
    if (A->n_bands <= 4 && spec == NMX_TOW_SPEC_C2) {
      hipLaunchKernelGGL((nmx_kern_timeosc_w1000_low<4, NMX_TOW_SPEC_C2>), dim3(grid), dim3(64), lds, s, *A, n_items);
      nmxi_note_kernel("nmx_kern_timeosc_w1000_low<4, 65809u>");
    } else if (A->n_bands <= 4 && spec == NMX_TOW_SPEC_DEFAULT) {
      hipLaunchKernelGGL((nmx_kern_timeosc_w1000_low<4, NMX_TOW_SPEC_DEFAULT>), dim3(grid), dim3(64), lds, s, *A, n_items);
      nmxi_note_kernel("nmx_kern_timeosc_w1000_low<4, 196915u>");
    } else if (A->n_bands <= 4) {
      hipLaunchKernelGGL(nmx_kern_timeosc_w1000_low<4>, dim3(grid), dim3(64), lds, s, *A, n_items);
      nmxi_note_kernel("nmx_kern_timeosc_w1000_low<4>");
    } else {
      hipLaunchKernelGGL(nmx_kern_timeosc_w1000_low<8>, dim3(grid), dim3(64), lds, s, *A, n_items);
      nmxi_note_kernel("nmx_kern_timeosc_w1000_low<8>");
    }
    return 1;
  }
  const size_t lds = (size_t)(A->stft.enabled ? NMX_TOW_LDS_FLOATS : NMX_TOW_LDS_FLOATS_NOSTFT) * 4;
  const int k = waves_per_wg(lds);
  const dim3 grid((unsigned)((n_items + k - 1) / k)), block(64 * k);
  const int slice = (int)(lds / 4);
  if (A->n_bands <= 4 && nmx_tow_spec(*A) == NMX_TOW_SPEC_ALL) {   // the headline set: feature tests folded
    hipLaunchKernelGGL((nmx_kern_timeosc_w1000<4, NMX_TOW_SPEC_ALL>), grid, block, lds * k, s, *A, n_items, slice);
    nmxi_note_kernel("nmx_kern_timeosc_w1000<4, 196923u>");
  } else if (A->n_bands <= 4) {
    hipLaunchKernelGGL(nmx_kern_timeosc_w1000<4>, grid, block, lds * k, s, *A, n_items, slice);
    nmxi_note_kernel("nmx_kern_timeosc_w1000<4>");
  } else {
    hipLaunchKernelGGL(nmx_kern_timeosc_w1000<8>, grid, block, lds * k, s, *A, n_items, slice);
    nmxi_note_kernel("nmx_kern_timeosc_w1000<8>");
  }
  return 1;
}

// The windows the matrix-pipe kernel (nmx_k_specmm.h) flagged -- a NaN or an infinity among the samples: it does not clean
// on load -- through the wave-level kernel's item code, which does.  One 16-bit mask per tile of that kernel (16
// consecutive windows of a channel, tile = group * n_channels + channel); a wave looks at 64 masks at a time; a clean
// recording never sets a bit.
template <int NB>
__global__ void __launch_bounds__(64) nmx_kern_timeosc_w1000_todo(const NmxTimeOscArgs A, int n_items) {
  const int lane = (int)(threadIdx.x & 63);
  const int C = A.n_channels, n_windows = n_items / C, n_tiles = ((n_windows + 15) / 16) * C;
  NmxW500TwReg T;
  bool loaded = false;
  for (int base = (int)blockIdx.x * 64; base < n_tiles; base += (int)gridDim.x * 64) {
    const int i = base + lane;
    const unsigned mine = i < n_tiles ? A.todo[i] : 0u;
    unsigned long long m = __ballot(mine != 0u);
    while (m) {
      const int src = (int)__ffsll((long long)m) - 1;
      m &= m - 1;
      const int tile = base + src;
      unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mine, src);
      const int g = tile / C, c = nmx_uniform_i(tile - g * C);
      while (bits) {
        const int j = __ffs((int)bits) - 1;
        bits &= bits - 1;
        if (!loaded) { T.load(A.w500_tab, lane); loaded = true; }
        const int w = nmx_uniform_i(16 * g + j);
        NmxTdRegs R;
        nmx_td_load<1000>(A, w, c, R);
        nmx_timeosc_w1000_body<NB, false, NmxW500TwReg, false, 0>(A, w, c, R, T, nmx_smem_wave);
        NMX_WAVE_FENCE();
      }
    }
  }
}
extern "C" void nmx_wave_launch_timeosc_w1000_todo(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  const size_t lds = (size_t)NMX_TOW_LDS_FLOATS_NOSTFT * 4;
  const int n_windows = n_items / A->n_channels, n_tiles = ((n_windows + 15) / 16) * A->n_channels;
  const int blocks = (n_tiles + 63) / 64;
  const int grid = blocks < 256 * 8 ? blocks : 256 * 8;
  if (A->n_bands <= 4) hipLaunchKernelGGL(nmx_kern_timeosc_w1000_todo<4>, dim3(grid), dim3(64), lds, s, *A, n_items);
  else hipLaunchKernelGGL(nmx_kern_timeosc_w1000_todo<8>, dim3(grid), dim3(64), lds, s, *A, n_items);
  nmxi_note_kernel("nmx_kern_timeosc_w1000_todo");
}

// STFT with 500-sample segments on windows of other lengths (<= 2048), the only time / oscillatory feature
template <int NB>
__global__ void __launch_bounds__(256) nmx_kern_timeosc_stft500(const NmxTimeOscArgs A, int n_items) {
  NMX_WAVE_ITEM(item, smem, n_items, NMX_TOS_LDS_FLOATS);
  nmx_timeosc_stft500_item<NB>(A, item / A.n_channels, item % A.n_channels, smem);
}
extern "C" int nmx_wave_launch_timeosc_stft500(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  if (!nmx_timeosc_stft500_ok(*A)) return 0;
  const int k = waves_per_wg((size_t)NMX_TOS_LDS_FLOATS * 4);
  const dim3 grid((unsigned)((n_items + k - 1) / k)), block(64 * k);
  if (A->n_bands <= 4) {
    hipLaunchKernelGGL(nmx_kern_timeosc_stft500<4>, grid, block, (size_t)NMX_TOS_LDS_FLOATS * 4 * k, s, *A, n_items);
    nmxi_note_kernel("nmx_kern_timeosc_stft500<4>");
  } else {
    hipLaunchKernelGGL(nmx_kern_timeosc_stft500<8>, grid, block, (size_t)NMX_TOS_LDS_FLOATS * 4 * k, s, *A, n_items);
    nmxi_note_kernel("nmx_kern_timeosc_stft500<8>");
  }
  return 1;
}

// 510-sample transforms (17 ms at 30 kHz): prime-factor transform per wave (nmx_k_timeosc_w510.h)
#ifndef NMX_W510_WAVES   // resident waves per SIMD the kernel is compiled for (4: 127 VGPRs, measured 6 % faster than 3)
#define NMX_W510_WAVES 4
#endif
#if NMX_W510_WAVES
#define NMX_W510_WPE __attribute__((amdgpu_waves_per_eu(NMX_W510_WAVES, NMX_W510_WAVES)))
#else
#define NMX_W510_WPE
#endif
template <int NB>
__global__ void __launch_bounds__(256) NMX_W510_WPE nmx_kern_timeosc_w510(const NmxTimeOscArgs A0, int n_items, int slice) {
  NMX_WAVE_ITEM(item, smem, n_items, slice);
  // (the plan through the kernel-argument segment pointer: the item launders it between its phases)
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* nmx_karg_p;
  const NmxTimeOscArgs& A = *(const NmxTimeOscArgs*)(nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  nmx_timeosc_w510_item<NB>(A, A.w510_tab, item / A.n_channels, item % A.n_channels, smem);
}
extern "C" int nmx_wave_launch_timeosc_w510(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  if (!nmx_timeosc_w510_ok(*A, A->w510_tab)) return 0;
  const int slice = NMX_TO510_LDS_FLOATS(A->W);
  const int k = waves_per_wg((size_t)slice * 4);
  const dim3 grid((unsigned)((n_items + k - 1) / k)), block(64 * k);
  if (A->n_bands <= 4) {
    hipLaunchKernelGGL(nmx_kern_timeosc_w510<4>, grid, block, (size_t)slice * 4 * k, s, *A, n_items, slice);
    nmxi_note_kernel("nmx_kern_timeosc_w510<4>");
  } else {
    hipLaunchKernelGGL(nmx_kern_timeosc_w510<8>, grid, block, (size_t)slice * 4 * k, s, *A, n_items, slice);
    nmxi_note_kernel("nmx_kern_timeosc_w510<8>");
  }
  return 1;
}

// register-resident scan (Hjorth / Raw / LineLength only): four waves per workgroup, no LDS
__global__ void __launch_bounds__(256) nmx_kern_scan(const NmxTimeOscArgs A, int n_items, int n_windows, int order) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  if (order) nmx_td_item(A, item % n_windows, item / n_windows);
  else nmx_td_item(A, item / A.n_channels, item % A.n_channels);
}

extern "C" void nmx_wave_launch_scan(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  const int order = 0;
  hipLaunchKernelGGL(nmx_kern_scan, dim3((n_items + 3) / 4), dim3(256), 0, s, *A, n_items, n_items / A->n_channels, order);
  nmxi_note_kernel("nmx_kern_scan");
}

// dense-first launch: compact LDS layout (more waves per CU); overflowing items are flagged
__global__ void __launch_bounds__(256) nmx_kern_sharp_dense(const NmxSharpArgs A, int n_items, int slice) {
  NMX_WAVE_ITEM(item, smem, n_items, slice);
  const int fi = item % A.n_filters, r = item / A.n_filters;
  nmx_sharp_item_dense(A, r / A.n_channels, r % A.n_channels, fi, (long long)item, smem);
}

extern "C" void nmx_wave_launch_sharp_dense(const NmxSharpArgs* A, int n_items, hipStream_t s) {
  const int slice = (A->dz_lds_floats + 3) & ~3;
  const int k = waves_per_wg((size_t)slice * 4, 2);
  hipLaunchKernelGGL(nmx_kern_sharp_dense, dim3((unsigned)((n_items + k - 1) / k)), dim3(64 * k), (size_t)slice * 4 * k, s, *A,
                     n_items, slice);
  nmxi_note_kernel("nmx_kern_sharp_dense");
}

// items the fused bank kernel could not finish (more than 128 extrema of a kind): generic list code.
// Persistent waves walk the flag array; on the bench workload no flag is ever set.
// (64 flags per wave and step -- one byte per lane, one ballot -- instead of one dependent load per flag: the scan of
// 524 288 clear flags took 0.3 ms of the side stream)
__global__ void __launch_bounds__(64) nmx_kern_sharp_todo(const NmxSharpArgs A, int n_items, const unsigned char* todo) {
  const int lane = (int)(threadIdx.x & 63);
  for (int base = (int)blockIdx.x * 64; base < n_items; base += (int)gridDim.x * 64) {
    const int i = base + lane;
    unsigned long long m = __ballot(i < n_items && todo[i] != 0);
    while (m) {
      const int item = base + (int)__ffsll((long long)m) - 1;
      m &= m - 1;
      const int fi = item % A.n_filters, r = item / A.n_filters;
      nmx_sharp_item(A, r / A.n_channels, r % A.n_channels, fi, nmx_smem_wave);
      NMX_SYNC();
    }
  }
}

extern "C" void nmx_wave_launch_sharp_todo(const NmxSharpArgs* A, int n_items, size_t lds, const unsigned char* todo,
                                           hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)nmx_kern_sharp_todo, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int blocks = (n_items + 63) / 64;
  const int grid = blocks < 256 * 14 ? blocks : 256 * 14;
  hipLaunchKernelGGL(nmx_kern_sharp_todo, dim3(grid), dim3(64), lds, s, *A, n_items, todo);
  nmxi_note_kernel("nmx_kern_sharp_todo");
}

extern "C" void nmx_wave_launch_burst_stat(const NmxBurstStatArgs* A, int n_items, size_t lds, hipStream_t s) {
  static unsigned long long seen = 0;
  if (nmx_first_on_device(seen)) {
    (void)hipFuncSetAttribute((const void*)nmx_kern_burst_stat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if ((A->W & 3) == 0 && A->W <= 2048) {
    if (A->W <= 1024) {
      hipLaunchKernelGGL(nmx_kern_burst_stat_reg<16>, dim3((n_items + 3) / 4), dim3(256), 0, s, *A, n_items);
      nmxi_note_kernel("nmx_kern_burst_stat_reg<16>");
    } else {
      hipLaunchKernelGGL(nmx_kern_burst_stat_reg<32>, dim3((n_items + 3) / 4), dim3(256), 0, s, *A, n_items);
      nmxi_note_kernel("nmx_kern_burst_stat_reg<32>");
    }
    return;
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
  const int k = waves_per_wg(lds, 2);
  const int slice = (int)((lds / 4 + 3) & ~(size_t)3);
  hipLaunchKernelGGL(nmx_kern_sharp, dim3((n_items + k - 1) / k), dim3(64 * k), (size_t)slice * 4 * k, s, *A,
                     n_items, slice);
  nmxi_note_kernel("nmx_kern_sharp");
}

// nmx_common.h -- shared host/device definitions of the nmx engine (gfx950 / CDNA4).
//
// Device code is written in "phase" style: every cooperative step is a loop
//     for (int j = NMX_TID; j < m; j += NMX_NT) { ... }   followed by NMX_SYNC()
// plus block reductions.  One workgroup processes one item ((channel, window) or
// (channel, band)); all staging happens in LDS.  Compiling with -DNMX_HOST_EMU maps
// NMX_TID/NMX_NT to 0/1 and NMX_SYNC to a no-op so the SAME source runs single-threaded
// under g++; that build exists only for tests/ (kernel-logic checks in a container that
// has no GPU) and is never part of libnmx.so -- the product has no CPU path.
#pragma once

#include <math.h>
#include <stdint.h>

#ifdef NMX_HOST_EMU
struct float2 {
  float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
#define NMX_DEV static inline
#define NMX_DEVM inline            /* member functions */
#define NMX_TID 0
#define NMX_NT 1
#define NMX_SYNC() ((void)0)
#define NMX_GLOBAL_FENCE() ((void)0)
#define NMX_RESTRICT
#else
#include <hip/hip_runtime.h>
#define NMX_DEV __device__ __forceinline__
#define NMX_DEVM __device__ __forceinline__
#ifdef NMX_NT_FIXED
// translation units of the one-item-per-WAVE kernels (nmx_wave.hip): the workgroup size is a
// compile-time 64, so no blockDim load from the implicit kernel arguments (a dependent global
// load at the top of every item) and every "block" reduction collapses to its wave part
#define NMX_TID ((int)(threadIdx.x & 63))
#define NMX_NT 64
#elif defined(NMX_BLOCK_FIXED)
// translation units whose kernels are always launched with NMX_BLOCK_FIXED threads: grid-stride loops
// over compile-time lengths get compile-time trip counts (less loop control on the scalar unit)
#define NMX_TID ((int)threadIdx.x)
#define NMX_NT NMX_BLOCK_FIXED
#else
#define NMX_TID ((int)threadIdx.x)
#define NMX_NT ((int)blockDim.x)
#endif
// Workgroup barrier.  A single-wave workgroup needs no s_barrier: its LDS operations execute
// in order, so a compiler fence + lgkmcnt(0) is enough -- and, unlike __syncthreads(), it does
// not drain vmcnt, i.e. it does not stall on outstanding global loads/stores (table prefetches,
// result stores) at every phase boundary.
#define NMX_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef NMX_NT_FIXED
#define NMX_SYNC() NMX_WAVE_FENCE()
#elif defined(NMX_BLOCK_FIXED)
#define NMX_SYNC()                                        \
  do {                                                    \
    if (NMX_BLOCK_FIXED <= 64) { NMX_WAVE_FENCE(); }      \
    else { __syncthreads(); }                             \
  } while (0)
#else
#define NMX_SYNC()                                   \
  do {                                               \
    if (blockDim.x <= 64) { NMX_WAVE_FENCE(); }      \
    else { __syncthreads(); }                        \
  } while (0)
#endif
#define NMX_RESTRICT __restrict__
// global-memory writes of this thread become visible to the other threads of the workgroup (followed by NMX_SYNC)
#define NMX_GLOBAL_FENCE() __threadfence_block()
// host-side helpers shared by the translation units of libnmx.so
// nmxi_note_kernel: every launcher records the kernel it actually launched (name as rocprofv3 prints it)
// under the current stage; nmx_last_kernels() reports them (bench.py's roofline names the kernel from here).
extern "C" void nmxi_note_kernel(const char* name);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE opt-in: true the first time the calling
// site runs on the current device (`seen` = a static bitmask of device ordinals owned by the call site)
static inline bool nmx_first_on_device(unsigned long long& seen) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
  if ((seen >> d) & 1ull) return false;
  seen |= 1ull << d;
  return true;
}
#endif

// Values that are wave-uniform by construction but that the compiler cannot prove uniform (results
// of vector memory loads, threadIdx >> 6): moving them to SGPRs keeps the derived addresses, buffer
// descriptors and branches on the scalar unit.
#ifdef NMX_HOST_EMU
static inline int nmx_uniform_i(int v) { return v; }
static inline long long nmx_uniform_ll(long long v) { return v; }
#else
__device__ __forceinline__ int nmx_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long nmx_uniform_ll(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
#endif

#define NMX_MAX_STAGES 12
#define NMX_MAX_BANDS_DEV 16
// table of the matrix-pipe spectrum kernel (nmx_k_specmm.h): granules of four n, n < 250 live
#define NMX_SMM_NG 63
#define NMX_SMM_TAB_FLOATS (4 * NMX_SMM_NG * 64)
#define NMX_MAX_FILTERS_DEV 24
#define NMX_MAX_SW_COMBOS_DEV 48

// ---- FFT plan (complex length n, mixed radix Stockham) --------------------------------
struct NmxFftStage {
  int radix;       // 2, 3, 4, 5 or any prime (generic O(p^2) butterfly)
  int ns;          // product of the radices of the previous stages
  int m;           // n / radix  (butterflies in this stage)
  unsigned magic;  // floor(2^32 / ns) + 1 : q = umulhi(j, magic) == j / ns for j < 2^16
  int tw_step;     // n / (ns * radix)
};

struct NmxFft {
  int n;
  int nstages;
  NmxFftStage st[NMX_MAX_STAGES];
  const float2* tw;   // [n]      exp(-2 pi i k / n)
  const float2* twr;  // [n + 1]  exp(-2 pi i k / (2 n)), real <-> half-length complex split
};

struct NmxCols {
  int base, ch_stride, a_stride, b_stride;
};

// One oscillatory family (FFT / Welch / STFT) as the kernel sees it
struct NmxOsc {
  int enabled;
  int n;         // real transform length (FFT: N, Welch/STFT: nperseg)
  int nfreq;     // n / 2 + 1
  int k_lo, k_hi;  // bins actually evaluated: union of the bands (all bins with return_spectrum)
  int nseg;      // segments (FFT: 1)
  int step;      // hop between segments
  int half;      // STFT: even-extension length nperseg / 2
  int complex_full;  // 1: n odd -> full-length complex transform of (x, 0)
  int log_transform;
  unsigned estimators;
  int n_est;
  int return_spectrum;
  float scale;   // Welch: 1 / (fs * sum w^2); STFT: 1 / sum w
  float log10_scale;   // log10(scale), formed on the host (the library log10f of a plan constant was ~40 VALU instructions per item)
  int bin_lo[NMX_MAX_BANDS_DEV], bin_hi[NMX_MAX_BANDS_DEV];
  float inv_bins[NMX_MAX_BANDS_DEV];   // 1 / (bin_hi - bin_lo), NaN for an empty band (the reference's mean of nothing)
  NmxCols cols, psd_cols;
  NmxFft fft;    // complex length n/2 (or n when complex_full)
  const float* win;  // [n] window (Welch: hann, STFT: hamming), NULL for FFT
  // STFT: the segments are transformed CENTRED (x - mean of the window: fp32 rounding relative to the signal, not to
  // its offset) and the constant's share comes back analytically: X_seg[k] += mean * wdc[sel][k], wdc[0][k] = DFT of
  // the window (a full segment: interior or even-extended edge), wdc[1][k] = DFT of its first n - nadd samples (the
  // zero-padded last segment of scipy's padded=True); float64 on the host, [2][nfreq] complex
  const float2* wdc;
  int nadd;          // STFT: zeros appended to the extended signal (the tail of the last segment)
};

struct NmxTimeOscArgs {
  int stft_per_wave;     // STFT segments distributed over the waves of the workgroup (needs 2 x 250 complex per buffer)
  const float* x;        // input samples
  long long ch_stride;   // elements between channels
  long long win_stride;  // elements between windows (0 for a strided stream view)
  const long long* starts;  // per-window start sample or NULL
  float* out;            // [n_windows][n_outputs]
  int n_outputs;
  int n_channels;
  int W;
  int n_bands;
  int clean_on_load;     // apply nan_to_num while loading
  unsigned features;     // NMX_F_* (HJORTH, RAW, LINELENGTH handled here)
  NmxCols hjorth_cols, raw_cols, ll_cols;
  NmxOsc fft, welch, stft;
  // LDS carve (float offsets)
  int off_x, off_a, off_b, off_spec, off_red, lds_floats;
  const float* w500_tab;   // W = 1000: tables of the wave-level kernel (nmx_k_fft500.h), else NULL
  const unsigned short* w510_tab;   // 510-sample transforms: position tables of the prime-factor wave kernel (nmx_k_timeosc_w510.h)
  const float* smm_tab;    // matrix-pipe spectrum kernel (nmx_k_specmm.h): [cos even k, cos odd k, sin even k, sin odd k][n / 4][row 16][n % 4], else NULL
  int smm_k0;              // first bin of that table
  int starts_mod4;         // every window start of this launch is a multiple of 4 samples (16-byte loads of the lanes' runs)
  const float* dcf;        // [n_channels] offset the windows were split from (nmx_engine_dc.inc), or NULL: the true window is
                           // x + dcf[c].  Differences and variances do not see a constant; Raw, bin 0 of the FFT and the STFT's
                           // window-DFT share (NmxOsc::wdc) do
  unsigned short* todo;    // matrix-pipe kernel: [ceil(n_windows / 16)][n_channels] masks, bit j = window 16 g + j of the channel holds
                           // a NaN / an infinity (the kernel does not clean on load) and is left to nmx_kern_timeosc_w1000_todo;
                           // NULL without that kernel
};

#define NMXD_F_HJORTH (1u << 0)
#define NMXD_F_RAW (1u << 1)
#define NMXD_F_BANDPOWER (1u << 2)
#define NMXD_F_STFT (1u << 3)
#define NMXD_F_FFT (1u << 4)
#define NMXD_F_WELCH (1u << 5)
#define NMXD_F_SHARPWAVE (1u << 6)
#define NMXD_F_BURSTS (1u << 7)
#define NMXD_F_LINELENGTH (1u << 8)

#define NMXD_EST_MEAN 1u
#define NMXD_EST_MEDIAN 2u
#define NMXD_EST_STD 4u
#define NMXD_EST_MAX 8u

// nmx_k_specmm.h -- FFT band power (+ Hjorth / LineLength / Raw) of 1000-sample windows with the spectrum on the MATRIX
// pipe: kernel A for BASELINE config[1] and bench.py's Mode A (features/oscillatory.py:90-119, hjorth_raw.py:24-42,
// linelength.py:11-21), the HBM-bound half of the north star.
//
// Why.  nmx_kern_timeosc_w1000_low sits on VALU issue (82 % of the slots; profiles/r03_pmc_step.log): ~290 instructions
// of time-domain statistics and ~220 of a 500-point transform per window, of which the band features read 31 BINS.  Those
// bins are a dense contraction,
//     X[k] = sum_n x[n] exp(-2 pi i k n / 1000),   k = k0 .. k0 + 31,
// i.e. (64 x 1000: cos and sin rows) x (1000 x windows).  On gfx950 `v_mfma_f32_32x32x2_f32` is EXACT fp32 (a k-ordered
// fmaf chain) at the fp32 vector rate -- no faster than the VALU, but it is a SEPARATE pipe: one instruction keeps it busy
// for 64 cycles while the wave's VALU slots stay free for the time-domain statistics.  The direct sum costs 128 kflop per
// window (the FFT 25), 0.85 ms per 1 M windows of otherwise idle matrix time.
//
// Mapping.  One wave = 32 windows (the MFMA's 32 columns), two accumulator tiles (cos rows, sin rows).  The instruction's
// two k slots are the two HALVES of the window: lane l streams samples n = 500 (l >> 5) + s of window l & 31, s = 0 .. 499,
// straight from global memory in 80-byte runs (its own lines: L1 keeps them for the five loads of a chunk), and reads the
// matching table entries T[row l & 31][500 (l >> 5) + s] from LDS, where the workgroup stages the table chunk by chunk
// (25 chunks of 20 samples per half, double buffered, one barrier per chunk; rows 80 bytes apart: conflict-free 16-byte
// reads).  No transpose, no shuffle: B operand = the sample the lane just loaded.
// Every sample a lane loads also advances ITS window's running statistics (single pass, shifted by the window's first
// sample: sum, squares, first and second differences, |d1|) on the VALU in the shadow of the MFMAs; the two halves of a
// window meet once, at the end (lane l <-> l + 32).
// The samples enter shifted by the pilot x[0] (sum_n exp(-2 pi i k n / 1000) = 0 for 1 <= k < 1000: the bins do not change,
// the fp32 sums no longer carry the offset).
// Conditions (host: nmx_specmm_ok): W = 1000, FFT over the whole window, band means only, every band inside 32 consecutive
// bins with k_lo >= 1, no Welch / STFT.  Device only.
#pragma once

#include "nmx_device.h"

#ifndef NMX_HOST_EMU
#include <type_traits>

#define NMX_SMM_CH 20                                   // samples per half and chunk
#define NMX_SMM_NCH 25                                  // 25 x 20 = 500
#define NMX_SMM_CHUNK_FLOATS (2 * 2 * 32 * NMX_SMM_CH)  // [table: cos, sin][half][row][20]
#define NMX_SMM_BUF_FLOATS (4 * 768)                    // a staged chunk: 640 float4 + the dead slots of the third staging round
#define NMX_SMM_LDS_FLOATS (2 * NMX_SMM_BUF_FLOATS)     // double buffered: 24 KiB
#define NMX_SMM_TAB_FLOATS (64 * 1000)                  // global table: [cos rows 0..31, sin rows 0..31][n]

typedef float nmx_v16 __attribute__((ext_vector_type(16)));
typedef float nmx_v4 __attribute__((ext_vector_type(4)));

static inline bool nmx_specmm_ok(const NmxTimeOscArgs& A) {
  if (!A.smm_tab || A.W != 1000 || A.n_bands > 8 || A.n_bands < 1) return false;
  if (A.welch.enabled || A.stft.enabled || !A.fft.enabled) return false;
  const NmxOsc& O = A.fft;
  if (O.complex_full || O.estimators != NMXD_EST_MEAN || O.return_spectrum || O.n != 1000) return false;
  if (!(O.k_lo >= 1 && O.k_hi - O.k_lo <= 32 && O.k_hi <= 500 && A.smm_k0 == O.k_lo)) return false;
  // the lanes read their runs with 16-byte loads
  return A.starts_mod4 && ((unsigned long long)A.x & 15ull) == 0 && (A.ch_stride & 3) == 0 && (A.win_stride & 3) == 0;
}

// per-lane running statistics of one half of a window (samples shifted by the pilot)
struct NmxSmmStat {
  float s0, q0, q1, q2, sa;       // sum u, sum u^2, sum d1^2, sum d2^2, sum |d1|
  float up, dp;                   // previous sample, previous first difference
  float u_first, u_second;        // first two samples of the run (the seam with the other half)
};

template <bool TD>
NMX_DEV void nmx_smm_sample(NmxSmmStat& S, float u, int idx /* compile-time position inside the run */) {
  if (!TD) return;
  S.s0 += u;
  S.q0 = fmaf(u, u, S.q0);
  if (idx == 0) { S.u_first = u; S.up = u; return; }
  const float d = u - S.up;
  S.sa += fabsf(d);
  S.q1 = fmaf(d, d, S.q1);
  if (idx == 1) { S.u_second = u; S.up = u; S.dp = d; return; }
  const float e = d - S.dp;
  S.q2 = fmaf(e, e, S.q2);
  S.up = u;
  S.dp = d;
}

// One tile = four groups (one per wave) of 32 CONSECUTIVE WINDOWS OF ONE CHANNEL: group g -> channel g % C, windows
// 32 (g / C) .. + 31.  The lanes of a wave then read one region of one row (overlapping hops share their cache lines; distinct
// windows lie 4 KB apart) -- 32 channels of one hop would be 32 rows, a recording length apart each: 64 DRAM pages and
// TLB entries per load instruction (measured: 1.85 ms per 1 M windows that way, matrix pipe 46 % busy).
template <int NB, bool TD, bool CLEAN, int RING>
NMX_DEV void nmx_specmm_tile(const NmxTimeOscArgs& A, long long group0, int n_windows, float* lds) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, j = lane & 31;
  const int C = A.n_channels;
  const long long g = group0 + wave;
  const int wb = (int)(g / C), c = (int)(g - (long long)wb * C);   // (wave-uniform; a group beyond the last one has wb past the end)
  int w = 32 * wb + j;
  const bool valid = w < n_windows;
  if (!valid) w = n_windows - 1;
#ifdef NMX_SMM_DEBUG_SAMEWIN   // timing experiment: every lane streams the group's first window (results are wrong)
  w = 32 * wb < n_windows ? 32 * wb : n_windows - 1;
#endif
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? A.starts[w] : 0ll) + 500 * half;
  const float* tab = A.smm_tab;

  nmx_v16 acc_c, acc_s;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_c[r] = 0.f; acc_s[r] = 0.f; }
  NmxSmmStat S;
  S.s0 = S.q0 = S.q1 = S.q2 = S.sa = S.up = S.dp = S.u_first = S.u_second = 0.f;

  // table staging: chunk ch of (table t, half h, row r) = tab[(32 t + r) * 1000 + 500 h + 20 ch + 0..19]: 5 float4;
  // 2 * 2 * 32 * 5 = 640 float4 per chunk over 256 threads
  auto tab_src = [&](int f, int ch) -> const nmx_v4* {   // f in [0, 640): f = 5 * seg + q, seg = (t * 2 + h) * 32 + r
    const int seg = f / 5, q = f - 5 * seg, t = seg >> 6, h = (seg >> 5) & 1, r = seg & 31;
    return (const nmx_v4*)(tab + (long long)(32 * t + r) * 1000 + 500 * h + NMX_SMM_CH * ch) + q;
  };
  // (every thread issues three loads -- the third one clamped, its store masked: a branch around a global load makes the
  // compiler's wait-count pass conservative at the join, and the chunk then waits for the loads it has just issued)
  nmx_v4 tg[3];
  auto tab_load = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int f = tid + 256 * i;
      tg[i] = *tab_src(f < 640 ? f : 639, ch);
    }
  };
  auto tab_store = [&](int buf) {   // (segment seg at 20 * seg floats: same order; slots 640 .. 767 are never read)
#pragma unroll
    for (int i = 0; i < 3; ++i) ((nmx_v4*)(lds + buf * NMX_SMM_BUF_FLOATS))[tid + 256 * i] = tg[i];
  };
  // the lanes' runs are prefetched RING - 1 chunks ahead (a ring of 80-byte buffers per lane): one chunk of matrix work is
  // ~1 us, a scattered 128-byte line from HBM under load takes several (measured with one chunk of distance: every chunk
  // waited for its loads, matrix pipe 50 % idle).  RING = 4 needs the 512-register budget (one workgroup per CU).
  nmx_v4 xb[RING][5];
  auto x_load = [&](int ch, nmx_v4* dst) {
#pragma unroll
    for (int q = 0; q < 5; ++q) dst[q] = ((const nmx_v4*)(src + NMX_SMM_CH * ch))[q];
  };

  x_load(0, xb[0]);
  x_load(1, xb[1]);
  if (RING == 4) x_load(2, xb[2]);
  tab_load(0);
  tab_store(0);
  __syncthreads();
  // pilot: the window's first sample (lane j of half 0 holds it)
  float pilot = xb[0][0].x;
  if (CLEAN) pilot = nmx_clean_bl(pilot);
  pilot = __shfl(pilot, j, 64);
  float last = 0.f;   // x[999] (Raw feature): the last sample of half 1

  // chunk ch: arithmetic on xc (= ring slot ch & 3) and LDS table buffer ch & 1; XPRE: prefetch the lanes' runs of chunk
  // ch + 3 into xn (= slot (ch + 3) & 3); TPRE: stage the table of chunk ch + 1
  auto chunk = [&](auto first_tag, auto xpre_tag, auto tpre_tag, int ch, const nmx_v4* xc, nmx_v4* xn) {
    constexpr bool FIRST = decltype(first_tag)::value, XPRE = decltype(xpre_tag)::value, TPRE = decltype(tpre_tag)::value;
    const int buf = ch & 1;
    if (XPRE) x_load(ch + RING - 1, xn);
    if (TPRE) tab_load(ch + 1);
    __builtin_amdgcn_sched_barrier(0);   // the prefetches are issued HERE, ahead of the chunk's arithmetic, not sunk below it
    const float* tl = lds + buf * NMX_SMM_BUF_FLOATS + (half * 32 + j) * NMX_SMM_CH;   // cos rows; sin rows 64 segments on
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const nmx_v4 ac = ((const nmx_v4*)tl)[q], as = ((const nmx_v4*)(tl + 64 * NMX_SMM_CH))[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = xc[q][i];
        if (CLEAN) v = nmx_clean_bl(v);
        const float u = v - pilot;
#ifdef NMX_SMM_DEBUG_HALFK   // timing experiment: half the matrix work on the same loads (results are wrong)
        if (i & 1) {
#endif
        acc_c = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i], u, acc_c, 0, 0, 0);
        acc_s = __builtin_amdgcn_mfma_f32_32x32x2f32(as[i], u, acc_s, 0, 0, 0);
#ifdef NMX_SMM_DEBUG_HALFK
        }
#endif
        nmx_smm_sample<TD>(S, u, FIRST ? 4 * q + i : 2);
        if (!TPRE && q == 4 && i == 3) last = v;   // (the last chunk is the one that stages no successor)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (TPRE) tab_store(buf ^ 1);
    __syncthreads();
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  if constexpr (RING == 4) {
    chunk(yes, yes, yes, 0, xb[0], xb[3]);
#pragma unroll 1
    for (int ch = 1; ch + 4 < NMX_SMM_NCH; ch += 4) {   // chunks 1 .. 20
      chunk(no, yes, yes, ch, xb[1], xb[0]);
      chunk(no, yes, yes, ch + 1, xb[2], xb[1]);
      chunk(no, yes, yes, ch + 2, xb[3], xb[2]);
      chunk(no, yes, yes, ch + 3, xb[0], xb[3]);
    }
    chunk(no, yes, yes, 21, xb[1], xb[0]);   // prefetches chunk 24
    chunk(no, no, yes, 22, xb[2], xb[1]);
    chunk(no, no, yes, 23, xb[3], xb[2]);
    chunk(no, no, no, 24, xb[0], xb[3]);
  } else {
    chunk(yes, yes, yes, 0, xb[0], xb[2]);
#pragma unroll 1
    for (int ch = 1; ch + 3 < NMX_SMM_NCH; ch += 3) {   // chunks 1 .. 21
      chunk(no, yes, yes, ch, xb[1], xb[0]);
      chunk(no, yes, yes, ch + 1, xb[2], xb[1]);
      chunk(no, yes, yes, ch + 2, xb[0], xb[2]);
    }
    chunk(no, yes, yes, 22, xb[1], xb[0]);   // prefetches chunk 24
    chunk(no, no, yes, 23, xb[2], xb[1]);
    chunk(no, no, no, 24, xb[0], xb[2]);
  }
  static_assert(NMX_SMM_NCH == 25, "the chunk schedule above is written out for 25 chunks");

  float* out_row = A.out + (long long)w * A.n_outputs;
  // ---- band means: row of accumulator register r = (r & 3) + 8 (r >> 2) + 4 half, bin k0 + row ---------------------
  {
    const NmxOsc& O = A.fft;
    float bs[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bs[b] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = O.k_lo + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float pw = acc_c[r] * acc_c[r] + acc_s[r] * acc_s[r];
      const float val = O.log_transform ? nmx_log10_half_fast(pw) : __builtin_amdgcn_sqrtf(pw);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (b < A.n_bands) bs[b] += (k >= O.bin_lo[b] && k < O.bin_hi[b]) ? val : 0.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= A.n_bands) continue;
      const float tot = bs[b] + __shfl_xor(bs[b], 32, 64);
      if (valid && half == 0) out_row[O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride] = tot * O.inv_bins[b];
    }
  }
  // ---- time domain: join the halves (half 1 continues half 0: n = 500 follows n = 499) --------------------------------
  if (TD) {
    // what half 1 needs from half 0: its last sample and last first difference; what the seam adds to half 1's sums
    const float up0 = __shfl(S.up, j, 64), dp0 = __shfl(S.dp, j, 64);
    float s0 = S.s0, q0 = S.q0, q1 = S.q1, q2 = S.q2, sa = S.sa;
    if (half == 1) {
      const float d500 = S.u_first - up0;            // x[500] - x[499]
      const float d501 = S.u_second - S.u_first;
      const float e500 = d500 - dp0, e501 = d501 - d500;
      sa += fabsf(d500);
      q1 = fmaf(d500, d500, q1);
      q2 = fmaf(e500, e500, fmaf(e501, e501, q2));
    }
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64); q1 += __shfl_xor(q1, 32, 64);
    q2 += __shfl_xor(q2, 32, 64); sa += __shfl_xor(sa, 32, 64);
    // telescoped sums of the differences (u = x - pilot: differences are those of x)
    const float u0 = __shfl(S.u_first, j, 64), u1 = __shfl(S.u_second, j, 64);          // x[0], x[1] (shifted)
    const float u999 = __shfl(S.up, j + 32, 64), d999 = __shfl(S.dp, j + 32, 64);
    const float xl = __shfl(last, j + 32, 64);
    if (valid && half == 0) {
      const float rW = 1.f / 1000.f, rW1 = 1.f / 999.f, rW2 = 1.f / 998.f;
      const float sd1 = u999 - u0, sd2 = d999 - (u1 - u0);
      const float v0 = (q0 - s0 * s0 * rW) * rW;
      const float v1 = (q1 - sd1 * sd1 * rW1) * rW1;
      const float v2 = (q2 - sd2 * sd2 * rW2) * rW2;
      if (A.features & NMXD_F_HJORTH) {
        float act, mob, comp;
        const bool normal = v0 > 1e-30f && v0 < 1e30f && v1 > 1e-30f && v1 < 1e30f && v2 < 1e30f && v2 >= 0.f;
        if (normal) {
          act = v0;
          mob = __builtin_amdgcn_sqrtf(v1 * __builtin_amdgcn_rcpf(v0));
          comp = __builtin_amdgcn_sqrtf(v2 * __builtin_amdgcn_rcpf(v1)) * __builtin_amdgcn_rcpf(mob);
        } else {   // flat / degenerate windows: the reference's nan_to_num placement on IEEE arithmetic
          const float a0 = v0 < 0.f ? 0.f : v0, a1 = v1 < 0.f ? 0.f : v1, a2 = v2 < 0.f ? 0.f : v2;
          act = nmx_clean(a0);
          mob = nmx_clean(sqrtf(a1 / a0));
          comp = nmx_clean(sqrtf(a2 / a1) / mob);
        }
        const int col = A.hjorth_cols.base + c * A.hjorth_cols.ch_stride;
        out_row[col] = act;
        out_row[col + A.hjorth_cols.a_stride] = mob;
        out_row[col + 2 * A.hjorth_cols.a_stride] = comp;
      }
      if (A.features & NMXD_F_LINELENGTH) out_row[A.ll_cols.base + c * A.ll_cols.ch_stride] = sa * rW1 * rW1;
      if (A.features & NMXD_F_RAW) out_row[A.raw_cols.base + c * A.raw_cols.ch_stride] = xl;
    }
  }
}
#endif

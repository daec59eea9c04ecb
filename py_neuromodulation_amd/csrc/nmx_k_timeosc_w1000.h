// nmx_k_timeosc_w1000.h -- kernel A for the default shape, ONE WAVE per (window, channel):
// W = 1000 samples, FFT over the whole window, Welch with one 1000-sample segment, STFT with five
// 500-sample segments (nperseg 500, hop 250, even boundary), band MEANS only (the default estimator).
// Any other configuration runs the generic workgroup kernel (nmx_k_timeosc.h); both follow the same
// reference arithmetic (features/hjorth_raw.py, linelength.py, oscillatory.py -- see nmx_k_timeosc.h).
//
// CDNA4 mapping
//   * the window is loaded ONCE with four range-checked 16-byte buffer loads per lane; Hjorth / LineLength
//     / Raw come straight from those registers (nmx_k_scan.h), then the window is parked in LDS (4 KB);
//   * every transform is the wave-level 500-point complex transform of nmx_k_fft500.h (radix 10.10.5,
//     17 twiddles per lane in VGPRs, wave-local fences, no workgroup barrier):
//       FFT + Welch  ONE transform of the centred window; the periodic hann window of Welch's single
//              segment is applied as a three-term convolution of that spectrum   (1 transform)
//       STFT   TWO real segments per transform, z = seg_a + i seg_b, separated afterwards with
//              A[k] = (Z[k] + conj Z[500-k]) / 2,  B[k] = (Z[k] - conj Z[500-k]) / 2i   (3 transforms)
//   * band means are accumulated per lane while the bins are produced (no spectrum buffer) and reduced
//     with DPP wave reductions.
// LDS per wave: xs[1000] + one in-place transform buffer of 500 complex = 8 KB.
#pragma once

#include "nmx_k_bank_w64.h"
#include "nmx_k_scan.h"
#include "nmx_k_td.h"

#ifndef NMX_HOST_EMU

// (every transform runs IN PLACE -- a single wave reads all the points of a stage before it writes any, nmx_k_fft500.h --
// so one 500-point buffer serves as input, ping and pong: 8 KB per wave with the STFT's copy of the window, 4 KB without)
#define NMX_TOW_LDS_FLOATS (1008 + 1000)
#define NMX_TOW_LDS_FLOATS_NOSTFT 1008

// can this configuration run on the wave kernel?  (host side, called by the launcher)
static inline bool nmx_timeosc_w1000_ok(const NmxTimeOscArgs& A) {
  if (!A.w500_tab || A.W != 1000 || A.n_bands > 8) return false;
  auto mean_only = [](const NmxOsc& O) {
    return !O.complex_full && O.estimators == NMXD_EST_MEAN && !O.return_spectrum;
  };
  if (A.fft.enabled && !(mean_only(A.fft) && A.fft.n == 1000)) return false;
  if (A.welch.enabled && !(mean_only(A.welch) && A.welch.n == 1000 && A.welch.nseg == 1)) return false;
  if (A.stft.enabled && !(mean_only(A.stft) && A.stft.n == 500 && A.stft.nseg == 5 && A.stft.step == 250 &&
                          A.stft.half == 250)) return false;
  return A.fft.enabled || A.welch.enabled || A.stft.enabled;
}

// the persistent, prefetching kernel (nmx_wave.hip: nmx_kern_timeosc_w1000_low) takes the configurations without an STFT
// whose bands all end below bin 100 -- the default four bands (4 - 35 Hz), BASELINE config[1]
static inline bool nmx_timeosc_w1000_low_ok(const NmxTimeOscArgs& A) {
  if (A.stft.enabled || !(A.fft.enabled || A.welch.enabled)) return false;
  if (A.fft.enabled && A.fft.k_hi > 100) return false;
  if (A.welch.enabled && A.welch.k_hi + 1 > 100) return false;
  return true;
}
// LDS of a wave in the low-band forms: ONE transform buffer -- a single wave reads all the points of a stage before it
// writes any, so the Stockham stages run in place -- and the 102 real-transform twiddles behind it
#define NMX_TOW_LOW_TWL_OFF 1008
#define NMX_TOW_LOW_LDS_FLOATS (1008 + 2 * 102)

// NB = compile-time bound on the number of bands (4 covers the default settings: half the select / add
// instructions per spectral value of the 8-band build)
template <int NB>
struct NmxBandAcc {
  float s[NB];
  NMX_DEV void clear() {
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = 0.f;
  }
  // value v of bin k joins every band whose [lo, hi) holds k
  NMX_DEV void add(const NmxOsc& O, int n_bands, int k, float v) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (b < n_bands) s[b] += (k >= O.bin_lo[b] && k < O.bin_hi[b]) ? v : 0.f;
  }
  // `rail`: the window holds samples on the rail (+-FLT_MAX: cleaned infinities, or what a re-reference makes of them).  Its
  // spectral magnitudes are of the size of the rail -- finite 300-odd logs in the reference's float64 --, their squares
  // overflow, and whether a transform's butterflies then leave +inf or inf - inf = NaN is its summation order: NaN is
  // reported as the overflow it is, +inf (an EMPTY band stays NaN: the reference's mean of nothing)
  NMX_DEV static float railed(float v, float inv, bool rail) { return (rail && v != v && inv == inv) ? INFINITY : v; }
  NMX_DEV void emit(const NmxOsc& O, int n_bands, int vals_per_bin, float* out_row, int c, int lane, bool rail = false) {
    if (NB == 4) {
      // the four band sums stay in lanes 0 / 16 / 32 / 48 (bands 0, 2, 1, 3): ONE multiplication and ONE store
      // instruction for the four results, no broadcast through scalar registers; 1 / bins from the plan (NaN for an
      // empty band: the reference's mean of nothing)
      const float r = nmx_wave_sum4_rows(s[0], s[1], s[2], s[3]);
      const int row = lane >> 4, b = ((row & 1) << 1) | (row >> 1);
      float inv = b == 0 ? O.inv_bins[0] : (b == 1 ? O.inv_bins[1] : (b == 2 ? O.inv_bins[2] : O.inv_bins[3]));
      if (vals_per_bin != 1) inv *= 1.f / (float)vals_per_bin;
      if ((lane & 15) == 0 && b < n_bands) out_row[O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride] = railed(r * inv, inv, rail);
      return;
    }
    float tot[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) tot[b] = s[b];
#pragma unroll
    for (int b = 0; b < NB; b += 4) nmx_wave_sum4(tot[b], tot[b + 1], tot[b + 2], tot[b + 3]);   // (NB is 4 or 8)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= n_bands) continue;
      if (vals_per_bin == 1) {   // (compile-time at the call sites: the plan's 1 / bins, NaN for an empty band)
        if (lane == 0) out_row[O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride] = railed(tot[b] * O.inv_bins[b], O.inv_bins[b], rail);
        continue;
      }
      const int cnt = (O.bin_hi[b] - O.bin_lo[b]) * vals_per_bin;
      if (lane == 0) out_row[O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride] = cnt > 0 ? railed(tot[b] * __builtin_amdgcn_rcpf((float)cnt), 1.f, rail) : NAN;
    }
  }
};

// Rt: the window of (w, c), loads issued by the caller (or in flight: the persistent kernel issues the loads of a wave's NEXT item
// before it works on the current one); T: the lane's twiddles.  LDS: fa[500] | fb[501] | xs[1000] (xs only with the STFT).
// LOW: the persistent kernel's form -- no STFT, every band below bin 100 (nmx_timeosc_w1000_low_ok), the first 102
// entries of the real-transform twiddle table in LDS at smem + NMX_TOW_LOW_TWL_OFF: NO vector-memory load inside the item (loads retire
// in order, so one would wait for the prefetch of the next window to land first).
// LOWNP: the low-band form WITHOUT the LDS twiddle copy (one item per workgroup: nothing to amortise a copy over).
// SPEC != 0: compiled for ONE feature set -- bits 0..8 the NMXD_F_* features, bit 16 / 17 the log_transform of the FFT /
// Welch family -- (nmx_tow_spec): the flags below are constants and their tests fold away.
#define NMX_TOW_SPEC_LOG_FFT (1u << 16)
#define NMX_TOW_SPEC_LOG_WELCH (1u << 17)
#define NMX_TOW_FEATS (NMXD_F_HJORTH | NMXD_F_RAW | NMXD_F_LINELENGTH | NMXD_F_FFT | NMXD_F_WELCH | NMXD_F_STFT)
// the two sets that get their own build of the persistent kernel: BASELINE config[1] (FFT band power + Hjorth +
// LineLength) and the time / oscillatory features of default_settings.yaml (+ Raw + Welch), log10 band powers
#define NMX_TOW_SPEC_C2 (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_FFT | NMX_TOW_SPEC_LOG_FFT)
#define NMX_TOW_SPEC_DEFAULT (NMXD_F_HJORTH | NMXD_F_RAW | NMXD_F_LINELENGTH | NMXD_F_FFT | NMXD_F_WELCH | NMX_TOW_SPEC_LOG_FFT | NMX_TOW_SPEC_LOG_WELCH)
// every time / oscillatory feature of the hot path, log10 band powers (bench.py's headline set)
#define NMX_TOW_SPEC_ALL (NMX_TOW_SPEC_DEFAULT | NMXD_F_STFT)
static inline unsigned nmx_tow_spec(const NmxTimeOscArgs& A) {
  return (A.features & NMX_TOW_FEATS) | (A.fft.enabled && A.fft.log_transform ? NMX_TOW_SPEC_LOG_FFT : 0u) |
         (A.welch.enabled && A.welch.log_transform ? NMX_TOW_SPEC_LOG_WELCH : 0u);
}
template <int NB, bool LOW = false, typename TW = NmxW500TwReg, bool LOWNP = false, unsigned SPEC = 0>
NMX_DEV void nmx_timeosc_w1000_body(const NmxTimeOscArgs& A, int w, int c, NmxTdRegs& Rt, const TW& T, float* smem) {
  const unsigned features = SPEC ? (SPEC & NMX_TOW_FEATS) : A.features;
  const bool fft_on = SPEC ? (SPEC & NMXD_F_FFT) != 0 : A.fft.enabled != 0;
  const bool welch_on = SPEC ? (SPEC & NMXD_F_WELCH) != 0 : A.welch.enabled != 0;
  const bool stft_on = !LOW && (SPEC ? (SPEC & NMXD_F_STFT) != 0 : A.stft.enabled != 0);
  const bool fft_log = SPEC ? (SPEC & NMX_TOW_SPEC_LOG_FFT) != 0 : A.fft.log_transform != 0;
  const bool welch_log = SPEC ? (SPEC & NMX_TOW_SPEC_LOG_WELCH) != 0 : A.welch.log_transform != 0;
  const int lane = (int)(threadIdx.x & 63);
  nmx_c2* fa = (nmx_c2*)smem;                    // [500]
  nmx_c2* fb = fa;                               // (the transforms run in place)
  float* xs = smem + 1008;                       // [1000] the window, natural order
  float* out_row = A.out + (long long)w * A.n_outputs;
  const int nb = A.n_bands;

  const bool spec1000 = fft_on || welch_on;
  float wsum;
  bool rail = false;   // (NmxBandAcc::emit)
  const float dcv = nmx_dc_of(A, c);   // the constant the window was split from (0 without a split)
  NMX_PROF_DECL
  {
    // time domain on packed arithmetic (nmx_k_td.h); it also leaves the centred window in fb for the transform
    const bool td = (features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) != 0;
    // (always: the window sum it forms is also the NaN / infinity test of the window)
    const bool fast = nmx_td_emit<1000, (!LOW || SPEC != 0), (SPEC & NMX_TOW_FEATS)>(A, w, c, Rt, spec1000 ? (float*)fb : nullptr);
    wsum = Rt.sum;
    rail = !fast;
    if (fast) {
      if (stft_on) {   // park the CENTRED window in LDS (group 3: lanes 0..57); see the STFT block below
        const float mean = wsum * (1.f / 1000.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < 3 || lane < 58) ((nmx_f4*)xs)[lane + 64 * k] = Rt.x[k] - mean;
      }
    } else {
      // a NaN or an infinity in the window: cleaning loads and the scalar formulation (nmx_k_scan.h)
      NmxScanRegs R;
      nmx_scan_load(A, w, c, R);
      R.sum = 0.f;
      if (td) {
        nmx_scan_emit(A, w, c, R);
      } else if (spec1000 || stft_on) {
        float p0 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) p0 += (R.x[k][0] + R.x[k][1]) + (R.x[k][2] + R.x[k][3]);   // out-of-range samples are 0
        R.sum = nmx_wave_reduce(p0, 0.f, [](float a_, float b_) { return a_ + b_; });
      }
      wsum = R.sum;
      const float mean = R.sum / 1000.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < 3 || lane < 58) {
          if (stft_on) ((nmx_f4*)xs)[lane + 64 * k] = nmx_f4{R.x[k][0] - mean, R.x[k][1] - mean, R.x[k][2] - mean, R.x[k][3] - mean};
          if (spec1000) ((nmx_f4*)fb)[lane + 64 * k] = nmx_f4{R.x[k][0] - mean, R.x[k][1] - mean, R.x[k][2] - mean, R.x[k][3] - mean};
        }
    }
  }
  NMX_WAVE_FENCE();
  NMX_PROF(0)

  NmxBandAcc<NB> acc;
  // ---- FFT and Welch share ONE transform: Z' = FFT_500 of the packed CENTRED window x - mean ----------
  //   * a constant only reaches bin 0, so for k >= 1 the spectrum X' of x - mean is the spectrum of x (what
  //     the FFT feature reads; with the DC offset gone before the fp32 transform it is also more accurate),
  //     and X[0] = sum(x);
  //   * Welch here is one segment = the window, constant detrend, PERIODIC hann w[n] = 1/2 - 1/2 cos(2 pi n / N):
  //     multiplying by w is a three-term convolution of the spectrum,
  //         Y[k] = X'[k] / 2 - (X'[k-1] + X'[k+1]) / 4,   X'[0] = 0,  X'[-k] = conj X'[k],
  //     so no second transform and no windowed copy of the window are needed.
  if (spec1000) {
    // bins read below: up to max(k_hi) (+1 for Welch's neighbours) and their mirror images 500 - k
    const int kmax = (fft_on ? A.fft.k_hi : 0) > (welch_on ? A.welch.k_hi + 1 : 0)
                         ? A.fft.k_hi : (welch_on ? A.welch.k_hi + 1 : 0);
    const float2* Z = (const float2*)((LOW || kmax <= 100) ? nmx_w500_fft_fwd_low(fb, fa, fb, T, lane, kmax)
                                                           : nmx_w500_fft<-1>(fb, fa, fb, T, lane));
    NMX_PROF(1)
    const float2* twr = (LOW && !LOWNP) ? (const float2*)(smem + NMX_TOW_LOW_TWL_OFF) : (fft_on ? A.fft : A.welch).fft.twr;
    auto xbin = [&](int k) -> float2 {   // X'[k], any k in [-1, 501]
      const int kk = k < 0 ? -k : (k > 500 ? 1000 - k : k);
      if (kk == 0) return make_float2(0.f, 0.f);
      const float2 X = nmx_rfft_bin(Z, twr, 500, kk);
      return (k < 0 || k > 500) ? make_float2(X.x, -X.y) : X;
    };
    if (fft_on) {
      const NmxOsc& O = A.fft;
      acc.clear();
      for (int k = O.k_lo + lane; k < O.k_hi; k += 64) {
        // (low-band forms: 0 <= k < 100, no mirrored bin)
        const float2 X = k == 0 ? make_float2(wsum + 1000.f * dcv, 0.f) : (LOW ? nmx_rfft_bin(Z, twr, 500, k) : xbin(k));
        const float pw = X.x * X.x + X.y * X.y;
        const float v = fft_log ? nmx_log10_half_fast(pw) : sqrtf(pw);   // log10 |X| = log10(|X|^2) / 2
        acc.add(O, nb, k, v);
      }
      acc.emit(O, nb, 1, out_row, c, lane, rail);
    }
    NMX_PROF(2)
#ifndef NMX_HOST_EMU
    if (welch_on && A.welch.k_lo >= 1 && A.welch.k_hi - A.welch.k_lo + 2 <= 64 && A.welch.k_hi <= 100) {
      // low bands, all bins in one pass: lane l forms X'[k_lo - 1 + l] ONCE and takes the two neighbours of the
      // three-term hann convolution from the lanes next to it (DPP) instead of two more bins from LDS
      const NmxOsc& O = A.welch;
      acc.clear();
      const int k = O.k_lo - 1 + lane;
      float2 X0 = nmx_rfft_bin(Z, twr, 500, k < 1 ? 1 : (k > 101 ? 101 : k));
      if (k < 1) X0 = make_float2(0.f, 0.f);   // X'[0] = 0: the constant went with the detrend
      auto shr1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); };
      auto shl1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); };
      const float2 Xm = make_float2(shr1(X0.x), shr1(X0.y)), Xp = make_float2(shl1(X0.x), shl1(X0.y));
      const float yr = 0.5f * X0.x - 0.25f * (Xm.x + Xp.x), yi = 0.5f * X0.y - 0.25f * (Xm.y + Xp.y);
      float p = (yr * yr + yi * yi) * (2.f * O.scale);   // (1 <= k < 500: two-sided density)
      if (welch_log) p = nmx_log10_fast(p);
      if (k >= O.k_lo && k < O.k_hi) acc.add(O, nb, k, p);
      acc.emit(O, nb, 1, out_row, c, lane, rail);
    } else
#endif
    if (welch_on) {
      const NmxOsc& O = A.welch;
      acc.clear();
      for (int k = O.k_lo + lane; k < O.k_hi; k += 64) {
        const float2 X0 = xbin(k), Xm = xbin(k - 1), Xp = xbin(k + 1);
        const float yr = 0.5f * X0.x - 0.25f * (Xm.x + Xp.x), yi = 0.5f * X0.y - 0.25f * (Xm.y + Xp.y);
        float p = (yr * yr + yi * yi) * O.scale;
        if (!(k == 0 || k == 500)) p *= 2.f;
        if (welch_log) p = nmx_log10_fast(p);
        acc.add(O, nb, k, p);
      }
      acc.emit(O, nb, 1, out_row, c, lane, rail);
    }
    NMX_WAVE_FENCE();
    NMX_PROF(3)
  }
#if defined(NMX_BANK_PROFILE) && !defined(NMX_HOST_EMU)
  if (w == 5 && (c == 3 || c == 40) && lane == 0)
    printf("timeosc profile w=%d c=%d: time domain %lld | transform %lld | fft bands %lld | welch bands %lld (cycles)\n", w, c,
           pf_acc[0], pf_acc[1], pf_acc[2], pf_acc[3]);
#endif
  // ---- STFT: segments 0..4 at extended positions 250 s .. 250 s + 499 (even extension by 250) -------
  // The segments are cut from the CENTRED window (xs holds x - mean): a hamming-weighted +-500 offset in every fp32
  // product and butterfly put the rounding of these bins at 1e-7 of the OFFSET (2 % of the entries of the headline
  // workload missed 1e-5 relative, the FFT / Welch features on their centred transform none).  What the constant
  // contributes to a segment's spectrum is known exactly: mean * DFT(window), NmxOsc::wdc -- for the periodic hamming
  // window 0.54 n at bin 0, -0.23 n at bin 1, nothing elsewhere -- and is added to the bins that read it.
  if (stft_on) {
    const NmxOsc& O = A.stft;
    const float mean = wsum * (1.f / 1000.f);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)O.win, 0, 2000, 0x00020000);
    float hw[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) hw[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, 4 * lane + 256 * q, 0, 0));
    acc.clear();
    const float lscale = O.log_transform ? O.log10_scale : 0.f;   // log10(|X| scale) = log10(|X|^2) / 2 + log10(scale)
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = lane + 64 * q;
        if (i < 500) {
          float va, vb;
          if (pr == 0) { va = xs[i < 250 ? 250 - i : i - 250]; vb = xs[i]; }              // segments 0, 1
          else if (pr == 1) { va = xs[i + 250]; vb = xs[i + 500]; }                        // segments 2, 3
          else { va = xs[i < 250 ? 750 + i : 1248 - i]; vb = 0.f; }                        // segment 4
          fb[i] = nmx_mk2(va * hw[q], vb * hw[q]);
        }
      }
      NMX_WAVE_FENCE();
      const nmx_c2* Z = O.k_hi <= 100 ? nmx_w500_fft_fwd_low(fb, fa, fb, T, lane, O.k_hi)
                                      : nmx_w500_fft<-1>(fb, fa, fb, T, lane);
      for (int k = O.k_lo + lane; k < O.k_hi; k += 64) {
        const nmx_c2 zk = Z[k == 500 ? 0 : k], zn = Z[k == 0 ? 0 : 500 - k];
        // A = (zk + conj zn) / 2,  B = -i (zk - conj zn) / 2
        float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y);
        float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x);
        if (k < 2) {   // (all five segments are full: wdc[0]; real for the symmetric window up to the table's rounding)
          const float2 t = O.wdc[k];
          const float mt = mean + dcv;   // (the window's own mean + the offset it was split from)
          ax += mt * t.x; ay += mt * t.y;
          bx += mt * t.x; by += mt * t.y;
        }
        const float pa = ax * ax + ay * ay;
        const float va = O.log_transform ? nmx_log10_half_fast(pa) + lscale : sqrtf(pa) * O.scale;
        acc.add(O, nb, k, va);
        if (pr < 2) {
          const float pb = bx * bx + by * by;
          const float vb = O.log_transform ? nmx_log10_half_fast(pb) + lscale : sqrtf(pb) * O.scale;
          acc.add(O, nb, k, vb);
        }
      }
      NMX_WAVE_FENCE();
    }
    acc.emit(O, nb, 5, out_row, c, lane, rail);
  }
}

// ---- STFT with 500-sample segments on windows of OTHER lengths (BASELINE config 3: 2000-sample windows at 2 kHz, nine
// segments -- the reference takes windowlength_ms as a sample count, features/oscillatory.py:199-213), one wave per
// (window, channel): the window in LDS, two real segments per wave-level 500-point transform as above.  Only when the
// STFT is the one time / oscillatory feature (the 256-thread generic kernel: 1.4-1.7 ms per 256 hops x 256 channels).
#define NMX_TOS_LDS_FLOATS (2048 + 1008)
static inline bool nmx_timeosc_stft500_ok(const NmxTimeOscArgs& A) {
  const NmxOsc& O = A.stft;
  if (!A.w500_tab || A.fft.enabled || A.welch.enabled || !O.enabled || A.n_bands > 8 || A.W > 2048 || A.W < 500 || (A.W & 3)) return false;
  if (A.features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) return false;
  return !O.complex_full && O.estimators == NMXD_EST_MEAN && !O.return_spectrum && O.n == 500 && O.step == 250 && O.half == 250 &&
         O.nseg >= 1 && O.nseg <= 16;
}
template <int NB>
NMX_DEV void nmx_timeosc_stft500_item(const NmxTimeOscArgs& A, int w, int c, float* smem) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const int lane = (int)(threadIdx.x & 63);
  const int W = A.W;
  float* xs = smem;                              // [W <= 2048]
  nmx_c2* fa = (nmx_c2*)(smem + 2048);           // [500]: the transforms run in place
  nmx_c2* fb = fa;
  float* out_row = A.out + (long long)w * A.n_outputs;
  const NmxOsc& O = A.stft;
  const int nb = A.n_bands;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4 * W, 0x00020000);
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  nmx_f4 xv[8];
  float psum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    xv[k] = nmx_f4{0.f, 0.f, 0.f, 0.f};
    if (256 * k >= W) break;
    const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane + 1024 * k, 0, 0);
    nmx_f4 v = {__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    if (A.clean_on_load) { v.x = nmx_clean_bl(v.x); v.y = nmx_clean_bl(v.y); v.z = nmx_clean_bl(v.z); v.w = nmx_clean_bl(v.w); }
    xv[k] = v;
    psum += (v.x + v.y) + (v.z + v.w);   // (samples beyond the row are 0: hardware range check)
  }
  // centred window (see nmx_timeosc_w1000_body: STFT)
  const float mean = nmx_wave_reduce(psum, 0.f, [](float a_, float b_) { return a_ + b_; }) / (float)W;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (256 * k >= W) break;
    if (4 * (lane + 64 * k) < W) ((nmx_f4*)xs)[lane + 64 * k] = xv[k] - mean;
  }
  NmxW500TwReg T;
  T.load(A.w500_tab, lane);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)O.win, 0, 2000, 0x00020000);
  float hw[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) hw[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, 4 * lane + 256 * q, 0, 0));
  NMX_WAVE_FENCE();
  auto xe = [&](int e) -> float {   // even extension by 250, zero padding beyond
    if (e < 250) return xs[250 - e];
    if (e < 250 + W) return xs[e - 250];
    if (e < 500 + W) return xs[W - 2 - (e - 250 - W)];
    return 0.f;
  };
  NmxBandAcc<NB> acc;
  acc.clear();
  const float lscale = O.log_transform ? O.log10_scale : 0.f;
  for (int sa = 0; sa < O.nseg; sa += 2) {
    const bool two = sa + 1 < O.nseg;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = lane + 64 * q;
      if (i < 500) fb[i] = nmx_mk2(xe(250 * sa + i) * hw[q], two ? xe(250 * (sa + 1) + i) * hw[q] : 0.f);
    }
    NMX_WAVE_FENCE();
    const nmx_c2* Z = O.k_hi <= 100 ? nmx_w500_fft_fwd_low(fb, fa, fb, T, lane, O.k_hi) : nmx_w500_fft<-1>(fb, fa, fb, T, lane);
    for (int k = O.k_lo + lane; k < O.k_hi; k += 64) {
      const nmx_c2 zk = Z[k == 500 ? 0 : k], zn = Z[k == 0 ? 0 : 500 - k];
      float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y);
      float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x);
      {   // the constant's share (the last segment may end in scipy's zero padding: second table)
        const float2 ta = O.wdc[((O.nadd && sa == O.nseg - 1) ? O.nfreq : 0) + k];
        const float2 tb = O.wdc[((O.nadd && sa + 1 == O.nseg - 1) ? O.nfreq : 0) + k];
        ax += mean * ta.x; ay += mean * ta.y;
        bx += mean * tb.x; by += mean * tb.y;
      }
      const float pa = ax * ax + ay * ay;
      acc.add(O, nb, k, O.log_transform ? nmx_log10_half_fast(pa) + lscale : sqrtf(pa) * O.scale);
      if (two) {
        const float pb = bx * bx + by * by;
        acc.add(O, nb, k, O.log_transform ? nmx_log10_half_fast(pb) + lscale : sqrtf(pb) * O.scale);
      }
    }
    NMX_WAVE_FENCE();
  }
  acc.emit(O, nb, O.nseg, out_row, c, lane);
}
#endif

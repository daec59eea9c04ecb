// nmx_k_timeosc.h -- kernel A: time-domain scan + FFT / Welch / STFT band power for one
// (window, channel) item per workgroup.
//
// Reference arithmetic reproduced (citations into /root/reference/py_neuromodulation):
//   features/hjorth_raw.py:24-42,51-57   Hjorth activity / mobility / complexity, Raw
//   features/linelength.py:11-21         sum|dx| / (W-1)^2
//   features/oscillatory.py:90-119       FFT: |rfft(x[-N:])| -> log10 -> estimators, bins [lo,hi)
//   features/oscillatory.py:150-182      Welch: hann, constant detrend, density, 50 % overlap
//   features/oscillatory.py:215-250      STFT: hamming, even boundary, spectrum scaling
// Data flow: window HBM -> LDS once (coalesced), everything else LDS/registers, ~35 floats
// written back per item.  Variances are two-pass (mean-shifted) in fp32.
#pragma once

#include "nmx_device.h"

// variance (ddof 0) of f(i), i in [0, n), where f reads LDS; two block reductions
template <typename F>
NMX_DEV float nmx_var(int n, float* red, F f) {
  float s = 0.f;
  for (int i = NMX_TID; i < n; i += NMX_NT) s += f(i);
  const float mean = nmx_block_sum(s, red) / (float)n;
  float q = 0.f;
  for (int i = NMX_TID; i < n; i += NMX_NT) {
    const float d = f(i) - mean;
    q += d * d;
  }
  return nmx_block_sum(q, red) / (float)n;
}

// radix-10 static plans (500 = 10 10 5: three LDS passes instead of four) measured slower here at
// 128 threads/item (3.8 vs 3.4 ms) and equal at 64: off
#ifndef NMX_TIMEOSC_R10
#define NMX_TIMEOSC_R10 false
#endif
NMX_DEV float nmx_nan_to_num(float v) { return nmx_clean(v); }
// band-pass activity cell: nan_to_num'd (bandpower.py:197) unless a Kalman scan follows, which needs
// the raw value (the reference filters before nan_to_num, bandpower.py:188-197)
NMX_DEV float nmx_bp_activity(float v, unsigned raw) { return raw ? v : nmx_clean(v); }

// Hjorth triple of a series y[0..n) in LDS, with the reference's nan_to_num placement.
// mode 0 = hjorth_raw.py (complexity divides by the nan_to_num'ed mobility),
// mode 1 = bandpower.py:185-207 (nan_to_num only on the final values).
NMX_DEV void nmx_hjorth(const float* y, int n, float* red, int mode, bool need_mc,
                        float& activity, float& mobility, float& complexity) {
  const float v0 = nmx_var(n, red, [&](int i) { return y[i]; });
  activity = v0;
  mobility = complexity = 0.f;
  if (!need_mc) return;
  const float v1 = nmx_var(n - 1, red, [&](int i) { return y[i + 1] - y[i]; });
  const float v2 = nmx_var(n - 2, red, [&](int i) { return (y[i + 2] - y[i + 1]) - (y[i + 1] - y[i]); });
  const float mob = sqrtf(v1 / v0);
  const float dmob = sqrtf(v2 / v1);
  if (mode == 0) {
    mobility = nmx_nan_to_num(mob);
    complexity = nmx_nan_to_num(dmob / mobility);
  } else {
    mobility = mob;
    complexity = dmob / mob;
  }
}

// `rail`: the window holds samples on the rail (+-FLT_MAX: cleaned infinities or what a re-reference makes of them).  Its
// spectral magnitudes are of the size of the rail, their squares overflow, and whether a transform's butterflies then leave
// +inf or inf - inf = NaN is its summation order: a NaN band mean / median / maximum is reported as the overflow it is,
// +inf -- as the wave-level kernels do (nmx_k_timeosc_w1000.h: NmxBandAcc::railed).  (An empty band stays NaN.)
NMX_DEV float nmx_railed(float v, bool rail) { return (rail && v != v) ? INFINITY : v; }
NMX_DEV void nmx_emit_bands(const NmxOsc& O, const float* spec, int vals_per_bin, int n_bands,
                            float* out_row, int c, float* red, bool rail = false) {
  if (O.estimators == NMXD_EST_MEAN && n_bands <= 8) {
    // default configuration: all band means with one multi-value reduction
    float p[8];
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
    for (int b = 0; b < 8; ++b) {
      float sacc = 0.f;
      if (b < n_bands) {
        const int cnt = (O.bin_hi[b] - O.bin_lo[b]) * vals_per_bin;
        const float* v = spec + (O.bin_lo[b] - O.k_lo) * vals_per_bin;
        for (int i = NMX_TID; i < cnt; i += NMX_NT) sacc += v[i];
      }
      p[b] = sacc;
    }
    nmx_block_sum_n<8>(p, red);
    if (NMX_TID == 0) {
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int b = 0; b < 8; ++b)
        if (b < n_bands) {
          const int cnt = (O.bin_hi[b] - O.bin_lo[b]) * vals_per_bin;
          out_row[O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride] = cnt > 0 ? nmx_railed(p[b] / (float)cnt, rail) : NAN;
        }
    }
    return;
  }
  for (int b = 0; b < n_bands; ++b) {
    const int lo = O.bin_lo[b], hi = O.bin_hi[b];
    const int cnt = (hi - lo) * vals_per_bin;
    const float* v = spec + (lo - O.k_lo) * vals_per_bin;
    int slot = 0;
    float mean = NAN;
    const bool need_mean = O.estimators & (NMXD_EST_MEAN | NMXD_EST_STD);
    if (need_mean && cnt > 0) mean = nmx_est_mean(v, cnt, red);
    const int col0 = O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride;
    if (O.estimators & NMXD_EST_MEAN) {
      if (NMX_TID == 0) out_row[col0 + slot * O.cols.b_stride] = cnt > 0 ? nmx_railed(mean, rail) : mean;
      ++slot;
    }
    if (O.estimators & NMXD_EST_MEDIAN) {
      const float r = cnt > 0 ? nmx_railed(nmx_est_median(v, cnt, red), rail) : NAN;
      if (NMX_TID == 0) out_row[col0 + slot * O.cols.b_stride] = r;
      ++slot;
    }
    if (O.estimators & NMXD_EST_STD) {
      const float r = cnt > 0 ? nmx_est_std(v, cnt, mean, red) : NAN;
      if (NMX_TID == 0) out_row[col0 + slot * O.cols.b_stride] = r;
      ++slot;
    }
    if (O.estimators & NMXD_EST_MAX) {
      const float r = cnt > 0 ? nmx_railed(nmx_est_max(v, cnt, red), rail) : NAN;
      if (NMX_TID == 0) out_row[col0 + slot * O.cols.b_stride] = r;
      ++slot;
    }
  }
}

// real transform of the packed / windowed segment already sitting in `bufB` (as n/2 complex,
// or n complex with zero imaginary part when O.complex_full); returns pointer to Z
NMX_DEV float2* nmx_osc_fft(const NmxOsc& O, float2* bufA, float2* bufB) {
  return nmx_fft_auto<-1, NMX_TIMEOSC_R10>(O.fft, bufB, bufA, bufB);
}

NMX_DEV float2 nmx_osc_bin(const NmxOsc& O, const float2* Z, int k) {
  if (O.complex_full) return Z[k];
  return nmx_rfft_bin(Z, O.fft.twr, O.fft.n, k);
}

NMX_DEV void nmx_time_osc_item(const NmxTimeOscArgs& A, int w, int c, float* smem) {
  float* xs = smem + A.off_x;
  float2* bufA = (float2*)(smem + A.off_a);
  float2* bufB = (float2*)(smem + A.off_b);
  float* spec = smem + A.off_spec;
  float* red = smem + A.off_red;
  const int W = A.W;
  float* out_row = A.out + (long long)w * A.n_outputs;
  const float dcv = A.dcf ? A.dcf[c] : 0.f;   // the constant the window was split from: window = xs + dcv

  // ---- stage the window in LDS (lane-consecutive, coalesced) ---------------------------
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? A.starts[w] : 0ll);
  {
    const int clean = A.clean_on_load;
    nmx_stage_row(src, W, [=](int i, float v) { xs[i] = clean ? nmx_clean(v) : v; });
  }
  NMX_SYNC();
  bool rail = false;   // samples on the rail in this window (nmx_emit_bands)
  if (A.fft.enabled || A.welch.enabled || A.stft.enabled) {
    int any = 0;
    for (int i = NMX_TID; i < W; i += NMX_NT) any |= !(fabsf(xs[i]) < 1e30f);
    rail = nmx_block_or(any, red) != 0;
    NMX_SYNC();
  }

  // ---- time-domain features: Hjorth + LineLength fused, two multi-value reductions ---------
  if (A.features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH)) {
    // pass 1: sums of x, dx, d2x (for the means) and of |dx| (line length)
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = NMX_TID; i < W; i += NMX_NT) {
      const float x0 = xs[i];
      p[0] += x0;
      if (i + 1 < W) {
        const float d1 = xs[i + 1] - x0;
        p[1] += d1;
        p[3] += fabsf(d1);
        if (i + 2 < W) p[2] += (xs[i + 2] - xs[i + 1]) - d1;
      }
    }
    nmx_block_sum_n<4>(p, red);
    const float m0 = p[0] / (float)W, m1 = p[1] / (float)(W - 1), m2 = p[2] / (float)(W - 2);
    const float ll = p[3];
    // pass 2: mean-shifted sums of squares (np.var is two-pass)
    float q[3] = {0.f, 0.f, 0.f};
    for (int i = NMX_TID; i < W; i += NMX_NT) {
      const float x0 = xs[i];
      const float e0 = x0 - m0;
      q[0] += e0 * e0;
      if (i + 1 < W) {
        const float d1 = xs[i + 1] - x0;
        const float e1 = d1 - m1;
        q[1] += e1 * e1;
        if (i + 2 < W) {
          const float e2 = ((xs[i + 2] - xs[i + 1]) - d1) - m2;
          q[2] += e2 * e2;
        }
      }
    }
    nmx_block_sum_n<3>(q, red);
    if (NMX_TID == 0) {
      if (A.features & NMXD_F_HJORTH) {
        const float v0 = q[0] / (float)W, v1 = q[1] / (float)(W - 1), v2 = q[2] / (float)(W - 2);
        // hjorth_raw.py:24-34: complexity divides by the nan_to_num'ed mobility
        const float mob = nmx_nan_to_num(sqrtf(v1 / v0));
        const float comp = nmx_nan_to_num(sqrtf(v2 / v1) / mob);
        const int col = A.hjorth_cols.base + c * A.hjorth_cols.ch_stride;
        out_row[col] = nmx_nan_to_num(v0);
        out_row[col + A.hjorth_cols.a_stride] = mob;
        out_row[col + 2 * A.hjorth_cols.a_stride] = comp;
      }
      if (A.features & NMXD_F_LINELENGTH) {
        const float wm1 = (float)(W - 1);
        out_row[A.ll_cols.base + c * A.ll_cols.ch_stride] = ll / wm1 / wm1;
      }
    }
  }
  if ((A.features & NMXD_F_RAW) && NMX_TID == 0)
    out_row[A.raw_cols.base + c * A.raw_cols.ch_stride] = xs[W - 1] + dcv;

  // ---- FFT band power -------------------------------------------------------------------
  if (A.fft.enabled) {
    const NmxOsc& O = A.fft;
    const int N = O.n, off = W - N;
    // the transform runs on x - mean: a constant only reaches bin 0 (X[0] = sum x), and without the offset the fp32
    // rounding of the transform is relative to the signal, not to its DC level
    float sm = 0.f;
    for (int i = NMX_TID; i < N; i += NMX_NT) sm += xs[off + i];
    const float xsum = nmx_block_sum(sm, red), mean = xsum / (float)N;
    NMX_SYNC();
    if (O.complex_full) {
      for (int i = NMX_TID; i < N; i += NMX_NT) bufB[i] = make_float2(xs[off + i] - mean, 0.f);
    } else {
      for (int i = NMX_TID; i < N / 2; i += NMX_NT)
        bufB[i] = make_float2(xs[off + 2 * i] - mean, xs[off + 2 * i + 1] - mean);
    }
    NMX_SYNC();
    const float2* Z = nmx_osc_fft(O, bufA, bufB);
    for (int k = O.k_lo + NMX_TID; k < O.k_hi; k += NMX_NT) {
      float2 X = nmx_osc_bin(O, Z, k);
      if (k == 0) X = make_float2(xsum + (float)N * dcv, 0.f);
      float v = nmx_sqrt_fast(X.x * X.x + X.y * X.y);
      if (O.log_transform) v = nmx_log10_fast(v);
      spec[k - O.k_lo] = v;
    }
    NMX_SYNC();
    nmx_emit_bands(O, spec, 1, A.n_bands, out_row, c, red, rail);
    if (O.return_spectrum)
      for (int k = NMX_TID; k < O.nfreq; k += NMX_NT)
        out_row[O.psd_cols.base + c * O.psd_cols.ch_stride + k * O.psd_cols.a_stride] = spec[k];
  }

  // ---- Welch ------------------------------------------------------------------------------
  if (A.welch.enabled) {
    const NmxOsc& O = A.welch;
    const int N = O.n;
    for (int sgi = 0; sgi < O.nseg; ++sgi) {
      const int s0 = sgi * O.step;
      float sm = 0.f;
      for (int i = NMX_TID; i < N; i += NMX_NT) sm += xs[s0 + i];
      const float mean = nmx_block_sum(sm, red) / (float)N;
      NMX_SYNC();
      if (O.complex_full) {
        for (int i = NMX_TID; i < N; i += NMX_NT)
          bufB[i] = make_float2((xs[s0 + i] - mean) * O.win[i], 0.f);
      } else {
        for (int i = NMX_TID; i < N / 2; i += NMX_NT)
          bufB[i] = make_float2((xs[s0 + 2 * i] - mean) * O.win[2 * i],
                                (xs[s0 + 2 * i + 1] - mean) * O.win[2 * i + 1]);
      }
      NMX_SYNC();
      const float2* Z = nmx_osc_fft(O, bufA, bufB);
      for (int k = O.k_lo + NMX_TID; k < O.k_hi; k += NMX_NT) {
        const float2 X = nmx_osc_bin(O, Z, k);
        float p = (X.x * X.x + X.y * X.y) * O.scale;
        const bool edge = (k == 0) || ((N % 2 == 0) && k == N / 2);
        if (!edge) p *= 2.f;
        spec[k - O.k_lo] = (sgi == 0) ? p : spec[k - O.k_lo] + p;
      }
      NMX_SYNC();
    }
    const float inv = 1.f / (float)O.nseg;
    for (int k = NMX_TID; k < O.k_hi - O.k_lo; k += NMX_NT) {
      float v = spec[k] * inv;
      if (O.log_transform) v = nmx_log10_fast(v);
      spec[k] = v;
    }
    NMX_SYNC();
    nmx_emit_bands(O, spec, 1, A.n_bands, out_row, c, red, rail);
    if (O.return_spectrum)
      for (int k = NMX_TID; k < O.nfreq; k += NMX_NT)
        out_row[O.psd_cols.base + c * O.psd_cols.ch_stride + k * O.psd_cols.a_stride] = spec[k];
  }

  // ---- STFT -----------------------------------------------------------------------------
  if (A.stft.enabled) {
    const NmxOsc& O = A.stft;
    const int N = O.n, h = O.half;
    // CENTRED segments (x - mean of the window; the zero padding stays zero); the constant's share of a segment's
    // spectrum is added back from the plan's float64 table (NmxOsc::wdc): fp32 rounding relative to the signal
    float smw = 0.f;
    for (int i = NMX_TID; i < W; i += NMX_NT) smw += xs[i];
    const float mean = nmx_block_sum(smw, red) / (float)W;
    NMX_SYNC();
    auto xe = [&](int e) -> float {  // even extension by h, zero padding beyond
      if (e < h) return xs[h - e] - mean;
      if (e < h + W) return xs[e - h] - mean;
      if (e < 2 * h + W) return xs[W - 2 - (e - h - W)] - mean;
      return 0.f;
    };
    const int seg_padded = O.nadd ? O.nseg - 1 : -1;
    auto with_dc = [&](float2 X, int sgi, int k) -> float2 {
      const float2 t = O.wdc[(sgi == seg_padded ? O.nfreq : 0) + k];
      return make_float2(X.x + (mean + dcv) * t.x, X.y + (mean + dcv) * t.y);
    };
    int sg_first = 0;
#ifndef NMX_HOST_EMU
    if (!O.complex_full && O.fft.n == 250 && NMX_NT >= 128 && A.stft_per_wave) {
      // default STFT (nperseg 500): every WAVE takes whole segments and runs its own 250-point
      // transform in its own slice of the buffers -- no workgroup barriers inside the segment loop and
      // 50 of 64 lanes busy in the radix-5 passes (a 128-thread workgroup had 50 of 128)
      const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), nwv = NMX_NT >> 6;
      float2* wA = bufA + wv * 250;
      float2* wB = bufB + wv * 250;
      NMX_SYNC();
      for (int sgi = wv; sgi < O.nseg; sgi += nwv) {
        const int s0 = sgi * O.step;
        for (int i = lane; i < 250; i += 64)
          wB[i] = make_float2(xe(s0 + 2 * i) * O.win[2 * i], xe(s0 + 2 * i + 1) * O.win[2 * i + 1]);
        NMX_WAVE_FENCE();
        const float2* Z = nmx_fft4_wave<-1, 250, 5, 5, 5, 2>(wB, wA, wB, O.fft.tw);
        for (int k = O.k_lo + lane; k < O.k_hi; k += 64) {
          const float2 X = with_dc(nmx_rfft_bin(Z, O.fft.twr, 250, k), sgi, k);
          float v = nmx_sqrt_fast(X.x * X.x + X.y * X.y) * O.scale;
          if (O.log_transform) v = nmx_log10_fast(v);
          spec[(k - O.k_lo) * O.nseg + sgi] = v;
        }
        NMX_WAVE_FENCE();
      }
      sg_first = O.nseg;
    }
#endif
    if (N <= 64 && sg_first == 0) {
      // short segments (e.g. 17 samples at 30 kHz: dozens of segments per window): one (segment, bin)
      // pair per thread, direct N-term DFT -- no transform buffers, no barrier per segment
      const int nb = O.k_hi - O.k_lo;
      NMX_SYNC();
      for (int idx = NMX_TID; idx < O.nseg * nb; idx += NMX_NT) {
        const int sgi = idx / nb, k = O.k_lo + (idx - sgi * nb);
        const int s0 = sgi * O.step;
        float re = 0.f, im = 0.f;
        for (int i = 0; i < N; ++i) {
          float sn, cs;
#ifdef NMX_HOST_EMU
          const double ang = -2.0 * 3.14159265358979323846 * (double)((k * i) % N) / (double)N;
          sn = (float)sin(ang); cs = (float)cos(ang);
#else
          sincospif(-2.f * (float)((k * i) % N) / (float)N, &sn, &cs);
#endif
          const float v = xe(s0 + i) * O.win[i];
          re += v * cs;
          im += v * sn;
        }
        const float2 Xd = with_dc(make_float2(re, im), sgi, k);
        re = Xd.x; im = Xd.y;
        float v = nmx_sqrt_fast(re * re + im * im) * O.scale;
        if (O.log_transform) v = nmx_log10_fast(v);
        spec[(k - O.k_lo) * O.nseg + sgi] = v;
      }
      sg_first = O.nseg;
    }
    for (int sgi = sg_first; sgi < O.nseg; ++sgi) {
      const int s0 = sgi * O.step;
      NMX_SYNC();
      if (O.complex_full) {
        for (int i = NMX_TID; i < N; i += NMX_NT) bufB[i] = make_float2(xe(s0 + i) * O.win[i], 0.f);
      } else {
        for (int i = NMX_TID; i < N / 2; i += NMX_NT)
          bufB[i] = make_float2(xe(s0 + 2 * i) * O.win[2 * i], xe(s0 + 2 * i + 1) * O.win[2 * i + 1]);
      }
      NMX_SYNC();
      const float2* Z = nmx_osc_fft(O, bufA, bufB);
      for (int k = O.k_lo + NMX_TID; k < O.k_hi; k += NMX_NT) {
        const float2 X = with_dc(nmx_osc_bin(O, Z, k), sgi, k);
        float v = nmx_sqrt_fast(X.x * X.x + X.y * X.y) * O.scale;
        if (O.log_transform) v = nmx_log10_fast(v);
        spec[(k - O.k_lo) * O.nseg + sgi] = v;
      }
    }
    NMX_SYNC();
    nmx_emit_bands(O, spec, O.nseg, A.n_bands, out_row, c, red, rail);
    if (O.return_spectrum)
      for (int k = NMX_TID; k < O.nfreq; k += NMX_NT) {
        float s = 0.f;
        for (int m = 0; m < O.nseg; ++m) s += spec[k * O.nseg + m];
        out_row[O.psd_cols.base + c * O.psd_cols.ch_stride + k * O.psd_cols.a_stride] =
            s / (float)O.nseg;
      }
  }
}

// nmx_k_resample.h -- raw_resampling per window (row a4 of the scope table).
//
// Reference: processing/resample.py:42-60 -> mne.filter.resample(x, up = new / old, down = 1)
// (FFT method, boxcar window, npad = "auto", pad = "reflect_limited").  MNE is absent from the
// reference tree; the algorithm is restated in oracle/mne_restated.py::resample (PARITY UNPINNED
// against MNE itself) and this kernel follows that restatement step by step:
//   1. odd-reflect ("reflect_limited") pad the window to n_pad = 2^ceil(log2(W + 2 min(W/8, 100)))
//   2. X = rfft(padded)                                  (half-length complex FFT + split)
//   3. keep / zero-extend to the n_new = round(ratio n_pad) point spectrum; the Nyquist bin of the
//      shorter length is doubled (down-sampling) or halved (up-sampling)
//   4. y = irfft(X, n_new) * ratio                      (half-length complex FFT; full when odd)
//   5. crop round(ratio pad_left) samples on the left, keep W_new = round(ratio W)
// One workgroup per (window, channel); everything between the HBM read of the raw window and the
// HBM write of the resampled one stays in LDS.
//
// LONG windows (poly_q > 0: the padded length n_pad = N does not fit one LDS transform -- an 8 - 44 kHz recording with 1 s
// windows brought to 1 kHz, the reference's default `raw_resampling`): steps 1 - 3 as a decimation-in-time sum over the
// q = N / m polyphase components x_r[j] = padded[q j + r] (m = poly_m complex points per transform),
//     X[k] = sum_r  w_N^(k r)  F_r[k mod m],      F_r = DFT_m(x_r),
// for the n_new / 2 + 1 bins the shorter length keeps -- two components per complex transform (x_r + i x_(r+1), one
// aligned 8-byte load per point), split by their Hermitian symmetry, w_N from a table of all N roots.  The padded window
// is never materialised: every pass reads the raw window (L2) through the reflection rule.  Steps 4 - 5 unchanged.
#pragma once

#include "nmx_device.h"

struct NmxResampleArgs {
  const float* x;            // raw stream / windows
  long long ch_stride, win_stride;
  const long long* starts;   // may be null
  float* y;                  // [n_windows][C][W_new]
  int n_channels;
  int W, W_new;              // raw / resampled window length
  int n_pad, pad_l;          // padded length (even), left pad
  int n_new, crop_l;         // resampled padded length, left crop
  int inv_full;              // n_new odd: full-length complex inverse
  float scale;               // ratio / n_new
  float nyq_scale;           // 2 (down) / 0.5 (up) applied to bin use_len / 2 when use_len is even
  int nyq_bin;               // -1: none
  int clean_on_load;
  NmxFft fwd;                // complex length n_pad / 2
  NmxFft inv;                // complex length n_new / 2 (n_new even) or n_new (odd)
  int off_x, off_a, off_b, off_X, lds_floats;
  int poly_q, poly_m;        // long windows: q polyphase components of m points (fwd = complex length m); 0: one transform
  const float2* tab_n;       // exp(-2 pi i j / n_pad), j = 0 .. n_pad - 1 (long windows only)
};

NMX_DEV void nmx_resample_item(const NmxResampleArgs& A, int w, int c, float* smem) {
  float* xs = smem + A.off_x;
  float2* bufA = (float2*)(smem + A.off_a);
  float2* bufB = (float2*)(smem + A.off_b);
  float2* X = (float2*)(smem + A.off_X);     // Hermitian half of the NEW spectrum, [n_new / 2 + 1]
  const int W = A.W, nh = A.n_pad >> 1;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? A.starts[w] : 0ll);
  if (A.poly_q > 0) {
    // ---- 1. - 3. for long windows: polyphase sum over the bins the new length keeps -----------
    const int clean = A.clean_on_load;
    const int q = A.poly_q, m = A.poly_m, pl = A.pad_l, nmask = A.n_pad - 1;
    auto raw = [&](int j) -> float { const float v = src[j]; return clean ? nmx_clean(v) : v; };
    const float x0 = raw(0), xl = raw(W - 1);
    auto ext = [&](int i) -> float {
      const int j = i - pl;
      if (j < 0) return (-j <= W - 1) ? 2.f * x0 - raw(-j) : 0.f;
      if (j < W) return raw(j);
      const int r = j - (W - 1);
      return (r <= W - 1) ? 2.f * xl - raw(W - 1 - r) : 0.f;
    };
    const int kmax_new = A.n_new >> 1, kuse = kmax_new < nh ? kmax_new : nh;
    for (int k = NMX_TID; k <= kmax_new; k += NMX_NT) X[k] = make_float2(0.f, 0.f);
    for (int r = 0; r < q; r += 2) {
      for (int j = NMX_TID; j < m; j += NMX_NT) bufB[j] = make_float2(ext(q * j + r), ext(q * j + r + 1));
      NMX_SYNC();
      const float2* Z = nmx_fft<-1>(A.fwd, bufB, bufA, bufB);
      for (int k = NMX_TID; k <= kuse; k += NMX_NT) {   // (bin k stays with its thread over the passes)
        const int kk = k & (m - 1);
        const float2 zk = Z[kk], zc = Z[(m - kk) & (m - 1)];
        const float2 f0 = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));   // F_r[k]
        const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));
        const float2 f1 = make_float2(d.y, -d.x);                                     // F_(r+1)[k]
        const float2 t0 = A.tab_n[(k * r) & nmask], t1 = A.tab_n[(k * (r + 1)) & nmask];
        X[k] = nmx_cadd(X[k], nmx_cadd(nmx_cmul(t0, f0), nmx_cmul(t1, f1)));
      }
      NMX_SYNC();
    }
    if (A.nyq_bin >= 0 && A.nyq_bin <= kmax_new && (A.nyq_bin % NMX_NT) == NMX_TID) {
      X[A.nyq_bin].x *= A.nyq_scale;
      X[A.nyq_bin].y *= A.nyq_scale;
    }
    NMX_SYNC();
  } else {
  {
    const int clean = A.clean_on_load;
    nmx_stage_row(src, W, [=](int i, float v) { xs[i] = clean ? nmx_clean(v) : v; });
  }
  NMX_SYNC();
  // ---- 1. reflect_limited padding, packed as n_pad / 2 complex samples --------------------
  {
    const float x0 = xs[0], xl = xs[W - 1];
    const int pl = A.pad_l;
    auto ext = [&](int i) -> float {
      const int j = i - pl;
      if (j < 0) return (-j <= W - 1) ? 2.f * x0 - xs[-j] : 0.f;
      if (j < W) return xs[j];
      const int r = j - (W - 1);              // 1, 2, ...: mirror of x[W - 1 - r]
      return (r <= W - 1) ? 2.f * xl - xs[W - 1 - r] : 0.f;
    };
    for (int i = NMX_TID; i < nh; i += NMX_NT) bufB[i] = make_float2(ext(2 * i), ext(2 * i + 1));
  }
  NMX_SYNC();
  // ---- 2./3. forward transform; bins of the new length ------------------------------------
  {
    const float2* Z = nmx_fft<-1>(A.fwd, bufB, bufA, bufB);
    const int kmax_new = A.n_new >> 1;        // irfft(X, n_new) reads bins 0 .. n_new / 2
    for (int k = NMX_TID; k <= kmax_new; k += NMX_NT) {
      float2 v = make_float2(0.f, 0.f);
      if (k <= nh) v = nmx_rfft_bin(Z, A.fwd.twr, nh, k);
      if (k == A.nyq_bin) { v.x *= A.nyq_scale; v.y *= A.nyq_scale; }
      X[k] = v;
    }
    NMX_SYNC();
  }
  }
  // ---- 4. inverse real transform of length n_new --------------------------------------------
  float* yout = A.y + ((long long)w * A.n_channels + c) * A.W_new;
  if (!A.inv_full) {
    const int mh = A.n_new >> 1;
    for (int k = NMX_TID; k < mh; k += NMX_NT) {
      float2 xk = X[k], xn = X[mh - k];
      // C2R semantics: the imaginary parts of the DC and Nyquist bins do not contribute
      if (k == 0) { xk.y = 0.f; xn.y = 0.f; }
      bufB[k] = nmx_irfft_pre(xk, xn, A.inv.twr[k]);
    }
    NMX_SYNC();
    const float* y = (const float*)nmx_fft<+1>(A.inv, bufB, bufA, bufB);
    for (int i = NMX_TID; i < A.W_new; i += NMX_NT) yout[i] = y[A.crop_l + i] * A.scale;
  } else {
    const int n = A.n_new;                   // odd: Hermitian extension, full complex inverse
    for (int k = NMX_TID; k < n; k += NMX_NT) {
      const int kk = k <= (n >> 1) ? k : n - k;
      float2 v = X[kk];
      if (k > (n >> 1)) v.y = -v.y;
      if (k == 0) v.y = 0.f;
      bufB[k] = v;
    }
    NMX_SYNC();
    const float2* y = nmx_fft<+1>(A.inv, bufB, bufA, bufB);
    for (int i = NMX_TID; i < A.W_new; i += NMX_NT) yout[i] = y[A.crop_l + i].x * A.scale;
  }
}

// nmx_k_bank_w64d.h -- FIR bank, circular-convolution length M = 1024, ONE WAVE per (window, PAIR of channels).
//
// The sibling of nmx_k_bank_w64c.h for windows of at most 512 samples: the reference's taps are as long as a second of
// signal (filter_length = sfreq - 1, filter/mne_filter.py:35-80), so only their central 2 W - 1 taps can touch a short
// window and the "same" part of the convolution needs M >= W + (W - 1) = 2 W - 1 -- 1023 for BASELINE config 5
// (512-sample windows at 30 kHz).  Two channels ride in the real and imaginary part of ONE 1024-point complex
// transform (real spectrum of the symmetric taps: the spectral step is a scaling); the transform is the
// register-blocked radix 16 - 16 - 4 one of nmx_k_bank_w64.h, which maps natural order in to the register order
// k = l + 64 t + 256 r (register 4 t + r) out in BOTH directions -- the spectrum goes back in after a compile-time
// register renaming, and the W <= 512 samples that are read are the registers r < 2 of the inverse (HALF).
// Before: these windows ran through the 1536-point channel-pair kernel (3.5 ms per 1024 hops x 512 channels).
// Device only; activity-only band power; every filter with W + (L' - 1) / 2 <= 1024 (L' = taps that touch the window).
#pragma once

#include "nmx_k_bank_w64c.h"
#include "nmx_k_bank_w64x2.h"

#if !defined(NMX_HOST_EMU) && defined(NMX_LDS_ASM)

#define NMX_W64D_H_FLOATS 1024   // per filter: [8][64] pairs (H[l + 64 (2 i)], H[l + 64 (2 i + 1)]), natural order

// HALF: W <= 512 -- only the output registers 4 t + r, r < 2 (samples l + 64 t + 256 r < 512) are formed
template <int HALF>
NMX_DEV void nmx_bank_w64d_item(const NmxBankW64Args& AA, int w, int c, float* smem, const float* tab) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const NmxBankArgs& A = AA.b;
  nmx_c2* X = (nmx_c2*)(smem + AA.off_X);
  const int W = A.W;
  const int l = (int)(threadIdx.x & 63);
  const bool two = c + 1 < A.n_channels;
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  const nmx_c2* twB = (const nmx_c2*)(tab + (size_t)A.n_filters * NMX_W64D_H_FLOATS);
  const nmx_c2* twC = twB + NMX_W64_TWB_N;
  nmx_c2 v[16], z[16];

  const nmx_rsrc r1 = nmx_make_rsrc(src, 4 * W);
  const nmx_rsrc r2 = nmx_make_rsrc(src + A.ch_stride, two ? 4 * W : 0);
  NMX_UNROLL
  for (int j = 0; j < 16; ++j) {
    if (HALF && j >= 8) { v[j] = nmx_mk2(0.f, 0.f); continue; }
    v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 4 * l + 256 * j, 0, 0));
    v[j].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 4 * l + 256 * j, 0, 0));
  }
  if (A.clean_on_load) {
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) v[j] = nmx_mk2(nmx_clean_bl(v[j].x), nmx_clean_bl(v[j].y));
  }
  // second channel at the first one's scale (exact power of two); an all-zero channel stays exactly zero
  float m1 = 0.f, m2 = 0.f;
  NMX_UNROLL
  for (int j = 0; j < 16; ++j) { m1 = fmaxf(m1, fabsf(v[j].x)); m2 = fmaxf(m2, fabsf(v[j].y)); }
  m1 = nmx_wave_reduce(m1, 0.f, [](float a_, float b_) { return fmaxf(a_, b_); });
  m2 = nmx_wave_reduce(m2, 0.f, [](float a_, float b_) { return fmaxf(a_, b_); });
  int e = 0;
  if (m1 > 0.f && m2 > 0.f) {
    e = __builtin_amdgcn_frexp_expf(m1) - __builtin_amdgcn_frexp_expf(m2);
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
  }
  NMX_UNROLL
  for (int j = 0; j < 16; ++j) v[j].y = __builtin_amdgcn_ldexpf(v[j].y, e);
  const nmx_c2 unscale = nmx_mk2(m1 > 0.f ? 1.f : 0.f, m2 > 0.f ? __builtin_amdgcn_ldexpf(1.f, -e) : 0.f);

  NMX_WSYNC();
  nmx_w64_fft1024<-1, 0>(v, X, twB, twC, l);
  NMX_UNROLL
  for (int i = 0; i < 16; ++i) z[i] = v[i];   // z[4 t + r] = Z[l + 64 t + 256 r]

  const int nf = A.n_filters;
  const unsigned h_addr = nmx_lds_addr(tab) + 8u * (unsigned)l;
  for (int fi = 0; fi < nf; ++fi) {
    const NmxFilterDev& F = A.f[fi];
    {   // spectral step into NATURAL register order: v[j] = H[l + 64 j] Z[l + 64 j], Z[l + 64 j] = z[NMX_J2I(j)]
      nmx_c2 h[8];
      nmx_ds_read_seq<512, 0>(h, h_addr + (unsigned)fi * (NMX_W64D_H_FLOATS * 4u), std::make_integer_sequence<int, 8>{});
      NMX_SCHED_FENCE();
      NMX_UNROLL
      for (int i = 0; i < 8; ++i) {
        v[2 * i] = nmx_pk_mul_lo(z[NMX_J2I(2 * i)], h[i]);
        v[2 * i + 1] = nmx_pk_mul_hi(z[NMX_J2I(2 * i + 1)], h[i]);
      }
    }
    nmx_w64_fft1024<+1, HALF>(v, X, twB, twC, l);   // v[4 t + r] = y[l + 64 t + 256 r]
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) {
      if (HALF && (i & 3) >= 2) continue;
      v[i] = v[i] * unscale;
    }
    if (F.bp_seglen > 0) {
      const unsigned span = (unsigned)F.bp_seglen;
      const int s_l = l - (W - F.bp_seglen);
      nmx_c2 acc = nmx_mk2(0.f, 0.f), acc2 = nmx_mk2(0.f, 0.f);
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        if (HALF && (i & 3) >= 2) continue;
        const float mk = (unsigned)(s_l + 64 * (i >> 2) + 256 * (i & 3)) < span ? 1.f : 0.f;
        const nmx_c2 val = v[i] * mk;
        acc = acc + val;
        acc2 = nmx_c2_fma(val, val, acc2);
      }
      const float inv_n = 1.f / (float)F.bp_seglen;
      auto add = [](float a_, float b_) { return a_ + b_; };
      float t1 = nmx_wave_reduce(acc.x, 0.f, add), t2 = nmx_wave_reduce(acc.y, 0.f, add);
      const float q1 = nmx_wave_reduce(acc2.x, 0.f, add), q2 = nmx_wave_reduce(acc2.y, 0.f, add);
      const float mean1 = t1 * inv_n, mean2 = t2 * inv_n;
      t1 = q1 - mean1 * t1;
      t2 = q2 - mean2 * t2;
      if (mean1 * mean1 * (float)F.bp_seglen > 4.f * t1 || mean2 * mean2 * (float)F.bp_seglen > 4.f * t2) {
        nmx_c2 a2 = nmx_mk2(0.f, 0.f);   // wave-uniform, rare: mean-shifted like np.var
        const nmx_c2 mm = nmx_mk2(mean1, mean2);
        NMX_UNROLL
        for (int i = 0; i < 16; ++i) {
          if (HALF && (i & 3) >= 2) continue;
          const float mk = (unsigned)(s_l + 64 * (i >> 2) + 256 * (i & 3)) < span ? 1.f : 0.f;
          const nmx_c2 d = (v[i] - mm) * mk;
          a2 = nmx_c2_fma(d, d, a2);
        }
        t1 = nmx_wave_reduce(a2.x, 0.f, add);
        t2 = nmx_wave_reduce(a2.y, 0.f, add);
      }
      if (l < 2 && (l == 0 || two)) {
        const float act = (l == 0 ? t1 : t2) * inv_n;
        const int col = A.bp_cols.base + (c + l) * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
        out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
      }
    }
    float* dsw = F.sw_index >= 0 ? A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W : nullptr;
    float* dyb = F.burst_index >= 0 ? AA.yb_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W : nullptr;
    for (int dst = 0; dst < 2; ++dst) {
      float* d = dst ? dyb : dsw;
      if (!d) continue;
      const long long next = (long long)(dst ? A.n_burst_bands : A.n_sw_filters) * W;
      const nmx_rsrc s1 = nmx_make_rsrc(d, 4 * W);
      const nmx_rsrc s2 = nmx_make_rsrc(d + next, two ? 4 * W : 0);
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        if (HALF && (i & 3) >= 2) continue;
        const int off = 4 * l + 256 * (i >> 2) + 1024 * (i & 3);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i].x), s1, off, 0, NMX_SERIES_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i].y), s2, off, 0, NMX_SERIES_STORE_AUX);
      }
    }
  }
}
#endif

// nmx_k_bank.h -- kernel B: per-(window, channel) FIR bank by FFT convolution in LDS.
//
// One forward real FFT of the (zero-padded or odd-reflected) window is shared by every
// filter; per filter: multiply by the REAL spectrum of the centred taps (zero-phase,
// symmetric, odd length -> real even spectrum, so no complex multiply and half the table),
// inverse real FFT, then the consumers run on the filtered series while it is still in LDS:
//   * BandPower tail statistics              features/bandpower.py:165-207
//   * Hilbert envelope -> HBM (Bursts)        features/bursts.py:153
//   * filtered series -> HBM (sharp waves, notch)  features/sharpwaves.py:242-251,
//                                                 filter/notch_filter.py:78-93
// "same" semantics of filter/mne_filter.py:110-126: y[n] = sum_k h[k] x[n + (L-1)/2 - k],
// x = 0 outside the window.  With circularly centred taps the wanted samples are outputs
// [0, W) of a length-M circular convolution, alias-free when M >= W + (L-1)/2.
// Notch (MNE _overlap_add_filter, phase="zero", pad="reflect_limited"): the same on the
// odd-reflected window; only (L-1)/2 reflected samples per side can reach the kept outputs,
// so the staged signal is W + (L-1) long and outputs are read at offset (L-1)/2.
//
// The intermediate (C, B, W) tensor of the reference never exists in HBM.
#pragma once

#include "nmx_k_timeosc.h"

#define NMX_UPS_SLOTS 512   // workgroups of a partitioned-mode launch (two per CU; one scratch slot each)

struct NmxFilterDev {
  const float* H;  // [M/2 + 1] real spectrum of centred taps, pre-scaled by 1/M (partitioned mode: [P][B + 1] complex)
  int half;        // (L - 1) / 2
  int bp_seglen;   // 0 = no BandPower epilogue
  int bp_band;
  int burst_index;  // -1 = none
  int sw_index;     // -1 = none
  int store_raw;    // notch: write y to yout[w][c][W]
};

struct NmxBankArgs {
  const float* x;
  long long ch_stride, win_stride;
  const long long* starts;
  float* out;
  int n_outputs, n_channels, W, clean_on_load;
  const float* dcf;  // [n_channels] constant ADDED to every sample on load, or NULL: the offset the engine carries next to a
                     // split stream (nmx_engine_dc.inc) -- a zero-padded FIR does see a window's DC level (edge transients)
  int M;            // circular convolution length (even)
  int pad_mode;     // 0 zero-pad ("same"), 1 odd reflection (notch)
  int n_edge;       // reflect_limited: samples available for reflection (min(L, W) - 1)
  int pad_half;     // pad_mode 1: (L - 1) / 2 staged on each side
  NmxFft fft;       // complex length M / 2
  int n_filters;
  NmxFilterDev f[NMX_MAX_FILTERS_DEV];
  unsigned bp_features;  // bit0 activity, bit1 mobility, bit2 complexity
  int bp_log;
  unsigned bp_kalman_mask;   // bit band: a Kalman scan follows -> store the raw (log-)activity
  NmxCols bp_cols;
  // Hilbert envelope (Bursts)
  int n_burst_bands;
  float* env_out;   // [n_windows][C][n_burst_bands][W]
  NmxFft hil_r;     // complex length W/2 (W even) or W (W odd): forward real transform
  NmxFft hil_c;     // complex length W: inverse of the one-sided spectrum
  int hil_full;     // W odd
  // filtered series to HBM
  int n_sw_filters;
  float* sw_out;    // [n_windows][C][n_sw_filters][W]
  float* y_out;     // notch: [n_windows][C][W]
  int residual;     // notch in residual form: the taps are g = delta - h and what is stored is x - g * x_ext.  The rounding
                    // of an fp32 FFT convolution is relative to what passes it: the whole signal for h, the few Hz around
                    // the line for g -- a tenth of it for broadband data -- and the STFT's near-null bins under log10 see
                    // the difference (headline: 0.82 % of the STFT entries beyond 1e-5 with h)
  // LDS carve (float offsets)
  int off_X, off_a, off_b, off_red, lds_floats;
  // PARTITIONED mode (partitioned != 0: windows x taps whose FFT convolution does not fit one LDS transform -- >= 6 kHz
  // recordings with 1 s windows): uniformly partitioned overlap-save, O((W + L) / B x L / B x B) where the definition
  // y[n] = sum_j h[j] x[n + half - j] costs O(W L).  With Hm = the longest half length and u[t] = x_ext[t - Hm],
  // t in [0, W + 2 Hm), filter f's output is y[n] = (h_f * u)[n + half_f + Hm].  Blocks of B samples, transforms of 2 B
  // real points (complex length B: `fft`):
  //   frame b  = u[(b - 1) B, (b + 1) B)          U_b = rfft(frame b)          computed ONCE per (window, channel)
  //   block b of h_f * u = last B samples of irfft(sum_p U_(b-p) H_(f,p)),      H_(f,p) = rfft(h_f[p B, (p + 1) B))
  // U_b of the item live in a scratch slot of the workgroup (global memory, L2 / MALL), the partition spectra
  // F.H = [P_f][B + 1] complex (pre-scaled by 1 / 2B) are shared by all items; y[W] in LDS (at off_X) for the epilogues;
  // burst bands leave as series (yb_out) for the stand-alone Hilbert kernel.  A workgroup walks items blockIdx.x,
  // blockIdx.x + gridDim.x, ... (one scratch slot per workgroup).
  int partitioned;
  float* yb_out;    // [n_windows][C][n_burst_bands][W]
  int ups_B, ups_frames, ups_hm;
  float2* ups_scratch;          // [gridDim.x][ups_frames][B + 1]
};

// partitioned overlap-save (NmxBankArgs::ups_*), step 1: the spectra of all frames of one (window, channel)
NMX_DEV void nmx_bank_ups_frames(const NmxBankArgs& A, const float* src, float2* S, float2* bufA, float2* bufB, float dc = 0.f) {
  const int W = A.W, B = A.ups_B, hm = A.ups_hm, ne = A.n_edge, clean = A.clean_on_load;
  auto raw = [&](int k) -> float { const float v = src[k]; return (clean ? nmx_clean(v) : v) + dc; };
  const float x0 = raw(0), xl = raw(W - 1);
  auto u = [&](int t) -> float {   // u[t] = x_ext[t - hm] on [0, W + 2 hm), 0 elsewhere
    if (t < 0 || t >= W + 2 * hm) return 0.f;
    const int k = t - hm;
    if (k >= 0 && k < W) return raw(k);
    if (A.pad_mode == 0) return 0.f;
    if (k < 0) return (-k <= ne) ? 2.f * x0 - raw(-k) : 0.f;   // odd reflection, reflect_limited (MNE _smart_pad)
    const int r = k - (W - 1);
    return (r <= ne) ? 2.f * xl - raw(W - 1 - r) : 0.f;
  };
  for (int b = 0; b < A.ups_frames; ++b) {
    const int t0 = (b - 1) * B;
    for (int i = NMX_TID; i < B; i += NMX_NT) bufB[i] = make_float2(u(t0 + 2 * i), u(t0 + 2 * i + 1));
    NMX_SYNC();
    const float2* Z = nmx_fft<-1>(A.fft, bufB, bufA, bufB);
    float2* Sb = S + (long long)b * (B + 1);
    for (int k = NMX_TID; k <= B; k += NMX_NT) Sb[k] = nmx_rfft_bin(Z, A.fft.twr, B, k);
    NMX_SYNC();
  }
  NMX_GLOBAL_FENCE();   // the frames are read back by other threads of the workgroup
  NMX_SYNC();
}

// step 2: filter F of the item into y[0..W) (LDS)
NMX_DEV void nmx_bank_ups_filter(const NmxBankArgs& A, const NmxFilterDev& F, const float2* S, float* y, float2* bufA,
                                 float2* bufB) {
  const int W = A.W, B = A.ups_B, Bh = B >> 1;
  const int L = 2 * F.half + 1, P = (L + B - 1) / B;
  const int off = F.half + A.ups_hm;   // y[n] = (h * u)[n + off]
  const float2* NMX_RESTRICT H = (const float2*)F.H;
  for (int b = off / B; b <= (off + W - 1) / B; ++b) {
    const int pmax = (P - 1 < b) ? P - 1 : b;   // frames below 0 are zero
    for (int k = NMX_TID; k <= Bh; k += NMX_NT) {
      const int k2 = B - k;
      float2 a1 = make_float2(0.f, 0.f), a2 = make_float2(0.f, 0.f);
      for (int p = 0; p <= pmax; ++p) {
        const float2* Sb = S + (long long)(b - p) * (B + 1);
        const float2* Hp = H + (long long)p * (B + 1);
        a1 = nmx_cadd(a1, nmx_cmul(Sb[k], Hp[k]));
        a2 = nmx_cadd(a2, nmx_cmul(Sb[k2], Hp[k2]));
      }
      bufB[k] = nmx_irfft_pre(a1, a2, A.fft.twr[k]);
      if (k2 != k && k2 < B) bufB[k2] = nmx_irfft_pre(a2, a1, A.fft.twr[k2]);
    }
    NMX_SYNC();
    const float* yf = (const float*)nmx_fft<+1>(A.fft, bufB, bufA, bufB);
    for (int i = NMX_TID; i < B; i += NMX_NT) {
      const int n = b * B + i - off;
      if (n >= 0 && n < W) y[n] = yf[B + i];
    }
    NMX_SYNC();
  }
}

NMX_DEV void nmx_bank_item(const NmxBankArgs& A, int w, int c, float* smem, int slot = 0) {
  float2* X = (float2*)(smem + A.off_X);
  float2* bufA = (float2*)(smem + A.off_a);
  float2* bufB = (float2*)(smem + A.off_b);
  float* red = smem + A.off_red;
  const int W = A.W, M = A.M, Mh = M >> 1;
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? A.starts[w] : 0ll);

  if (A.partitioned) {   // ---- partitioned overlap-save (see NmxBankArgs::partitioned): y in LDS, the same epilogues ----------
    float* y = smem + A.off_X;
    float2* S = A.ups_scratch + (long long)slot * A.ups_frames * (A.ups_B + 1);
    nmx_bank_ups_frames(A, src, S, bufA, bufB, A.dcf ? A.dcf[c] : 0.f);
    for (int fi = 0; fi < A.n_filters; ++fi) {
      const NmxFilterDev& F = A.f[fi];
      nmx_bank_ups_filter(A, F, S, y, bufA, bufB);
      NMX_SYNC();
      if (F.bp_seglen > 0) {
        const bool need_mc = (A.bp_features & 6u) != 0;
        float act, mob, comp;
        nmx_hjorth(y + (W - F.bp_seglen), F.bp_seglen, red, 1, need_mc, act, mob, comp);
        if (NMX_TID == 0) {
          int col = A.bp_cols.base + c * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
          if (A.bp_features & 1u) {
            out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
            col += A.bp_cols.b_stride;
          }
          if (A.bp_features & 2u) { out_row[col] = nmx_nan_to_num(mob); col += A.bp_cols.b_stride; }
          if (A.bp_features & 4u) out_row[col] = nmx_nan_to_num(comp);
        }
      }
      float* dsts[3] = {
          F.sw_index >= 0 ? A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W : nullptr,
          F.store_raw ? A.y_out + ((long long)w * A.n_channels + c) * W : nullptr,
          F.burst_index >= 0 ? A.yb_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W : nullptr};
      for (int d = 0; d < 3; ++d)
        if (dsts[d]) {
          if (d == 1 && A.residual)
            for (int i = NMX_TID; i < W; i += NMX_NT) dsts[d][i] = (A.clean_on_load ? nmx_clean(src[i]) : src[i]) - y[i];
          else
            for (int i = NMX_TID; i < W; i += NMX_NT) dsts[d][i] = y[i];
        }
      NMX_SYNC();
    }
    return;
  }
  // ---- stage the (padded) window as M/2 packed complex samples in bufB -----------------
  if (A.pad_mode == 0) {
    for (int i = NMX_TID; i < Mh; i += NMX_NT) {
      const int n0 = 2 * i;
      float v0 = n0 < W ? src[n0] : 0.f, v1 = (n0 + 1) < W ? src[n0 + 1] : 0.f;
      if (A.clean_on_load) {
        v0 = nmx_clean(v0);
        v1 = nmx_clean(v1);
      }
      if (A.dcf) {
        if (n0 < W) v0 += A.dcf[c];
        if (n0 + 1 < W) v1 += A.dcf[c];
      }
      bufB[i] = make_float2(v0, v1);
    }
  } else {
    float* xs = (float*)X;  // the spectrum region doubles as window staging
    for (int i = NMX_TID; i < W; i += NMX_NT) {
      float v = src[i];
      if (A.clean_on_load) v = nmx_clean(v);
      if (A.dcf) v += A.dcf[c];
      xs[i] = v;
    }
    NMX_SYNC();
    const int h = A.pad_half, ne = A.n_edge;
    const float x0 = xs[0], xl = xs[W - 1];
    auto ext = [&](int jp) -> float {  // jp in [0, W + 2h): sample of the reflected signal
      const int j = jp - h;
      if (j < 0) return (-j <= ne) ? 2.f * x0 - xs[-j] : 0.f;
      if (j < W) return xs[j];
      const int r = j - (W - 1);
      return (r <= ne && j < W + h) ? 2.f * xl - xs[W - 1 - r] : 0.f;
    };
    for (int i = NMX_TID; i < Mh; i += NMX_NT) {
      const int n0 = 2 * i;
      bufB[i] = make_float2(n0 < W + 2 * h ? ext(n0) : 0.f, (n0 + 1) < W + 2 * h ? ext(n0 + 1) : 0.f);
    }
  }
  NMX_SYNC();

  // ---- forward transform, Hermitian half X[0..M/2] kept in LDS ------------------------
  {
    const float2* Z = nmx_fft<-1>(A.fft, bufB, bufA, bufB);
    for (int k = NMX_TID; k <= Mh; k += NMX_NT) X[k] = nmx_rfft_bin(Z, A.fft.twr, Mh, k);
    NMX_SYNC();
  }

  const int yoff = (A.pad_mode == 1) ? A.pad_half : 0;
  for (int fi = 0; fi < A.n_filters; ++fi) {
    const NmxFilterDev& F = A.f[fi];
    const float* NMX_RESTRICT H = F.H;
    // Z'[k] from X[k] H[k] and X[M/2 - k] H[M/2 - k]
    for (int k = NMX_TID; k < Mh; k += NMX_NT) {
      const float hk = H[k], hn = H[Mh - k];
      const float2 xk = X[k], xn = X[Mh - k];
      bufB[k] = nmx_irfft_pre(make_float2(xk.x * hk, xk.y * hk), make_float2(xn.x * hn, xn.y * hn),
                              A.fft.twr[k]);
    }
    NMX_SYNC();
    float2* yz = nmx_fft<+1>(A.fft, bufB, bufA, bufB);
    float2* other = (yz == bufA) ? bufB : bufA;
    const float* y = (const float*)yz + yoff;  // y[n], n in [0, W), natural order

    if (F.bp_seglen > 0) {  // ---- BandPower (bandpower.py:185-207) ----
      const bool need_mc = (A.bp_features & 6u) != 0;
      float act, mob, comp;
      nmx_hjorth(y + (W - F.bp_seglen), F.bp_seglen, red, 1, need_mc, act, mob, comp);
      if (NMX_TID == 0) {
        int col = A.bp_cols.base + c * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
        if (A.bp_features & 1u) {
          out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
          col += A.bp_cols.b_stride;
        }
        if (A.bp_features & 2u) {
          out_row[col] = nmx_nan_to_num(mob);
          col += A.bp_cols.b_stride;
        }
        if (A.bp_features & 4u) out_row[col] = nmx_nan_to_num(comp);
      }
    }
    if (F.sw_index >= 0) {
      float* dst = A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W;
      for (int i = NMX_TID; i < W; i += NMX_NT) dst[i] = y[i];
    }
    if (F.store_raw) {
      float* dst = A.y_out + ((long long)w * A.n_channels + c) * W;
      if (A.residual)
        for (int i = NMX_TID; i < W; i += NMX_NT) dst[i] = (A.clean_on_load ? nmx_clean(src[i]) : src[i]) - y[i];
      else
        for (int i = NMX_TID; i < W; i += NMX_NT) dst[i] = y[i];
    }
    if (F.burst_index >= 0) {  // ---- |hilbert(y)| (bursts.py:153), exact length-W transform ----
      const int Wh = W >> 1;
      const float2* Zy;
      if (A.hil_full) {
        for (int i = NMX_TID; i < W; i += NMX_NT) other[i] = make_float2(y[i], 0.f);
        NMX_SYNC();
        Zy = nmx_fft<-1>(A.hil_r, other, yz, other);
      } else {
        // y is contiguous floats: reinterpret as W/2 packed complex samples (read-only input)
        Zy = nmx_fft<-1>(A.hil_r, (const float2*)y, other, yz);
      }
      float2* Ab = (Zy == bufA) ? bufB : bufA;
      // one-sided spectrum: X[0], 2 X[k] (0 < k < W/2), X[W/2] (W even), 0 for negative k
      const float invW = 1.f / (float)W;
      for (int k = NMX_TID; k < W; k += NMX_NT) {
        float2 v = make_float2(0.f, 0.f);
        if (A.hil_full) {
          if (k == 0) v = Zy[0];
          else if (k <= (W - 1) / 2) v = make_float2(2.f * Zy[k].x, 2.f * Zy[k].y);
        } else if (k <= Wh) {
          v = nmx_rfft_bin(Zy, A.hil_r.twr, Wh, k);
          if (k != 0 && k != Wh) v = make_float2(2.f * v.x, 2.f * v.y);
        }
        Ab[k] = make_float2(v.x * invW, v.y * invW);
      }
      NMX_SYNC();
      float2* Zbuf = (Ab == bufA) ? bufB : bufA;
      const float2* an = nmx_fft<+1>(A.hil_c, Ab, Zbuf, Ab);
      float* dst = A.env_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W;
      for (int i = NMX_TID; i < W; i += NMX_NT) dst[i] = nmx_sqrt_fast(an[i].x * an[i].x + an[i].y * an[i].y);
    }
    NMX_SYNC();
  }
}

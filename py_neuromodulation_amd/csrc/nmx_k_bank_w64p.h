// nmx_k_bank_w64p.h -- the FIR-bank item of the PERSISTENT kernel, software-pipelined by hand.
//
// Same arithmetic, operation for operation, as nmx_bank_w64_item<0, 1, 0, 0, 0, HALF, 1> (nmx_k_bank_w64.h) --
// the results are bit-identical (tests: batch == one-window call) -- but the wave no longer sits idle while
// its LDS traffic drains.  Per-phase s_memtime counters of the straight-line version (-DNMX_BANK_PROFILE)
// showed, per filter: ~600 cycles waiting for the A/B table values in four dependent groups, ~1000 cycles
// draining the 16 pass-B stores before pass C may read, ~1400 cycles in the variance epilogue and ~400 in the
// series stores -- with 2 waves per SIMD there is nobody to fill those holes.  Here
//   * the (A_k, B_k) pairs of filter f + 1 are fetched (one unpaired ds_read_b64 per point, interleaved
//     table) while filter f runs pass C and its epilogue -- they are in registers when the spectral step starts;
//   * the epilogue of filter f (tail variance -> activity; series -> HBM) is DEFERRED: its VALU work runs
//     after the pass-A stores of filter f + 1 have been issued, its HBM stores after the pass-B stores -- the
//     exchange tile drains underneath.  The deferred outputs cost 16 VGPRs (windows <= 1024 samples: HALF).
// Device only; windows with an even number of samples (8-byte row accesses), activity-only band power.
#pragma once

#include "nmx_k_bank_w64.h"

#if !defined(NMX_HOST_EMU) && defined(NMX_LDS_ASM)

template <int HALF>
NMX_DEV void nmx_bank_w64_item_pipe(const NmxBankW64Args& AA, int w, int c, float* smem, const float* tab) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const NmxBankArgs& A = AA.b;
  nmx_c2* X = (nmx_c2*)(smem + AA.off_X);
  const int W = A.W;
  const int l = (int)(threadIdx.x & 63);
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  const nmx_c2* twB = (const nmx_c2*)(tab + (size_t)A.n_filters * 2 * NMX_W64_N);
  const nmx_c2* twC = twB + NMX_W64_TWB_N;
  constexpr int NO = HALF ? 8 : 16;           // output registers that exist
  nmx_c2 v[16], zr[16], zcr[16], hab[16], yp[NO];
  NMX_PROF_DECL

  // (A_k, B_k) of filter 0: in flight during the forward transform
  const unsigned tab_addr = nmx_lds_addr(tab) + 8u * (unsigned)l;
  nmx_ds_read_seq<512, 0>(hab, tab_addr, std::make_integer_sequence<int, 16>{});

  // ---- forward transform (as the generic item) -------------------------------------------------
  {
    const nmx_rsrc rs = nmx_make_rsrc(src, 4 * W);
    NMX_UNROLL
    for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, 8 * l + 512 * r, 0, 0);
    if (A.clean_on_load) {
      NMX_UNROLL
      for (int r = 0; r < 16; ++r) v[r] = nmx_mk2(nmx_clean_bl(v[r].x), nmx_clean_bl(v[r].y));
    }
  }
  NMX_WSYNC();
  nmx_w64_passA<-1>(v, X, l);
  NMX_WSYNC();
  nmx_w64_passB_load_lds<-1>(v, X, twB, l);
  NMX_WSYNC();
  nmx_w64_passB_store(v, X, l);
  NMX_WSYNC();
  nmx_w64_passC_lds<-1>(v, X, twC, l);
  NMX_UNROLL
  for (int i = 0; i < 16; ++i) zr[i] = v[i];
  NMX_WSYNC();
  NMX_PROF(0)
  NMX_UNROLL
  for (int r = 0; r < 16; ++r) {   // conjugate partners Z[n - k], once per item
    const nmx_c2 zs = zr[NMX_J2I(15 - r)];
    nmx_c2 zc = nmx_mk2(__shfl(zs.x, (64 - l) & 63), __shfl(zs.y, (64 - l) & 63));
    if (l == 0) zc = (r == 0) ? zr[0] : zr[NMX_J2I((16 - r) & 15)];
    zcr[r] = zc;
  }
  NMX_PROF(6)

  // deferred epilogue, part 1: tail variance of the band-pass filter `fe` -> activity (VALU + one 4-byte store)
  auto variance = [&](int fe) {
    const NmxFilterDev& F = A.f[fe];
    if (F.bp_seglen <= 0) return;
    const int lo = W - F.bp_seglen, hi = W;
    nmx_c2 acc = nmx_mk2(0.f, 0.f), acc2 = nmx_mk2(0.f, 0.f);
    const unsigned span = (unsigned)(hi - lo);
    const int s_l = 2 * l - lo;
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) {
      if (HALF && (i & 3) >= 2) continue;
      const int sb = s_l + 2 * (64 * (i >> 2) + 256 * (i & 3));
      nmx_c2 val = yp[HALF ? 2 * (i >> 2) + (i & 3) : i];
      val.x = (unsigned)sb < span ? val.x : 0.f;
      val.y = (unsigned)(sb + 1) < span ? val.y : 0.f;
      acc = nmx_cadd(acc, val);
      acc2 = nmx_c2_fma(val, val, acc2);
    }
    float tot = nmx_wave_reduce(acc.x + acc.y, 0.f, [](float a_, float b_) { return a_ + b_; });
    const float tot2 = nmx_wave_reduce(acc2.x + acc2.y, 0.f, [](float a_, float b_) { return a_ + b_; });
    const float mean = tot / (float)F.bp_seglen;
    tot = tot2 - mean * tot;
    if (mean * mean * (float)F.bp_seglen > 4.f * tot) {   // wave-uniform, rare: mean-shifted redo (np.var)
      nmx_c2 a2 = nmx_mk2(0.f, 0.f);
      const nmx_c2 mean2 = nmx_mk2(mean, mean);
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        if (HALF && (i & 3) >= 2) continue;
        const int sb = s_l + 2 * (64 * (i >> 2) + 256 * (i & 3));
        nmx_c2 d = nmx_csub(yp[HALF ? 2 * (i >> 2) + (i & 3) : i], mean2);
        d.x = (unsigned)sb < span ? d.x : 0.f;
        d.y = (unsigned)(sb + 1) < span ? d.y : 0.f;
        a2 = nmx_c2_fma(d, d, a2);
      }
      tot = nmx_wave_reduce(a2.x + a2.y, 0.f, [](float a_, float b_) { return a_ + b_; });
    }
    const float act = tot / (float)F.bp_seglen;
    if (l == 0) {
      const int col = A.bp_cols.base + c * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
      out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
    }
  };
  // deferred epilogue, part 2: filtered series of filter `fe` -> HBM (sharp-wave / burst hand-off)
  auto series = [&](int fe) {
    const NmxFilterDev& F = A.f[fe];
    float* dsw = F.sw_index >= 0 ? A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W : nullptr;
    float* dyb = F.burst_index >= 0 ? AA.yb_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W : nullptr;
    for (int dst = 0; dst < 2; ++dst) {
      float* d = dst ? dyb : dsw;
      if (!d) continue;
      const nmx_rsrc rs = nmx_make_rsrc(d, 4 * W);
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        if (HALF && (i & 3) >= 2) continue;
        __builtin_amdgcn_raw_buffer_store_b64(yp[HALF ? 2 * (i >> 2) + (i & 3) : i], rs,
                                              8 * l + 512 * (i >> 2) + 2048 * (i & 3), 0, 0);
      }
    }
  };

  const int nf = A.n_filters;
  for (int fi = 0; fi < nf; ++fi) {
    // ---- spectral step from registers: Z'[k] = A_k Z[k] + i B_k conj(Z[n-k]) -----------------------
    nmx_lds_wait8(hab); nmx_lds_tie8(hab + 8);
    NMX_UNROLL
    for (int r = 0; r < 16; ++r) v[r] = nmx_axpby_swap_pair(hab[r], zr[NMX_J2I(r)], zcr[r]);
    nmx_w64_passA<+1>(v, X, l);
    if (fi > 0) variance(fi - 1);            // the pass-A stores drain underneath
    NMX_WSYNC();
    NMX_PROF(1)
    nmx_w64_passB_load_lds<+1>(v, X, twB, l);
    NMX_PROF(2)
    nmx_w64_passB_store(v, X, l);
    if (fi > 0) series(fi - 1);              // the pass-B stores drain underneath
    NMX_WSYNC();
    NMX_PROF(3)
    if (HALF) nmx_w64_passC_lds_half<+1>(v, X, twC, l);
    else nmx_w64_passC_lds<+1>(v, X, twC, l);
    if (fi + 1 < nf)                          // (A_k, B_k) of the next filter: in flight during the epilogue
      nmx_ds_read_seq<512, 0>(hab, tab_addr + (unsigned)(fi + 1) * (2u * NMX_W64_N * 4u), std::make_integer_sequence<int, 16>{});
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) {
      if (HALF && (i & 3) >= 2) continue;
      yp[HALF ? 2 * (i >> 2) + (i & 3) : i] = v[i];
    }
    NMX_PROF(4)
  }
  variance(nf - 1);
  NMX_PROF(5)
  series(nf - 1);
  NMX_WSYNC();
  NMX_PROF(7)
  NMX_PROF_PRINT(w, c)
}
#endif

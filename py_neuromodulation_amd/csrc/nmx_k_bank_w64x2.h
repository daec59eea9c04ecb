// nmx_k_bank_w64x2.h -- FIR bank with a circular-convolution length of M = 4096 (half-length complex transform
// n = 2048), ONE WAVE per (window, channel): windows + filter half-length in (2048, 4096] -- BASELINE config 3:
// 2 kHz, 2000-sample windows, 1999 taps.
//
// The 2048-point transform is TWO of the register-blocked 1024-point transforms of nmx_k_bank_w64.h (passes A, B, C
// through the wave's one exchange tile, one after the other) plus an in-lane radix-2 step:
//   forward (decimation in time):  u[j] = (x[2j], x[2j+1]);  E = FFT_1024(u[2m]),  O = FFT_1024(u[2m+1]);
//            Z[k] = E[k] + w^k O[k],  Z[k + 1024] = E[k] - w^k O[k],  w = exp(-2 pi i / 2048)
//            -- a lane loads (u[2m], u[2m+1]) = x[4m .. 4m+3] as ONE 16-byte access, and both 1024-point results
//            leave pass C in the same registers-to-index map, so the combine needs no data movement;
//   spectral step as in the 1024-point kernel, Z'[k] = A_k Z[k] + i B_k conj(Z[2048 - k]): the partner of a "low"
//            point k < 1024 is the "high" point 1024 - k of the mirrored lane and vice versa;
//   inverse (decimation in frequency):  Ye[k] = Z'[k] + Z'[k + 1024],  Yo[k] = (Z'[k] - Z'[k + 1024]) conj(w^k);
//            y[2m] = IFFT_1024(Ye)[m], y[2m+1] = IFFT_1024(Yo)[m] -- the lane ends with x'[4m .. 4m+3]: 16-byte stores.
// LDS per workgroup: (A_k, B_k) tables of the first `n_tab_lds` filters (16 KiB each: as many as fit next to the
// tiles; the others are read from L2), the pass B / C twiddles, w^k (8 KiB), one 8.5 KiB exchange tile per wave.
// Device only (unpaired LDS reads, explicit operand modifiers); windows with W % 4 == 0, activity-only band power.
#pragma once

#include "nmx_k_bank_w64.h"

#if !defined(NMX_HOST_EMU) && defined(NMX_LDS_ASM)

typedef float nmx_f4 __attribute__((ext_vector_type(4)));
typedef unsigned nmx_u4 __attribute__((ext_vector_type(4)));

// natural butterfly index r = t + 4 r' of output register i = 4 t + r' (inverse of NMX_J2I)
#define NMX_I2J(i) (((i) >> 2) + 4 * ((i) & 3))

// one 1024-point transform through the tile: v[r] = X[lane + 64 r] in, v[4 t + r'] = Y[lane + 64 t + 256 r'] out
template <int DIR, int HALF>
NMX_DEV void nmx_w64_fft1024(nmx_c2* v, nmx_c2* X, const nmx_c2* twB, const nmx_c2* twC, int l) {
  nmx_w64_passA<DIR>(v, X, l);
  NMX_WSYNC();
  nmx_w64_passB_load_lds<DIR>(v, X, twB, l);
  nmx_w64_passB_store(v, X, l);
  NMX_WSYNC();
  if (HALF) nmx_w64_passC_lds_half<DIR>(v, X, twC, l);
  else nmx_w64_passC_lds<DIR>(v, X, twC, l);
  NMX_WSYNC();   // (the tile is free again: the next transform's pass A stores may follow)
}

// HALF: W <= 2048 -- only the outputs m < 512 of the inverse transforms (registers 4 t + r', r' < 2) are formed
template <int HALF>
NMX_DEV void nmx_bank_w64x2_item(const NmxBankW64Args& AA, int w, int c, float* smem, const float* tab, int n_tab_lds) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const NmxBankArgs& A = AA.b;
  nmx_c2* X = (nmx_c2*)(smem + AA.off_X);
  const int W = A.W;
  const int l = (int)(threadIdx.x & 63);
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  const float* tw_base = tab + (size_t)n_tab_lds * 4096;   // after the LDS-resident filter tables (2048 pairs = 4096 floats each)
  const nmx_c2* twB = (const nmx_c2*)tw_base;
  const nmx_c2* twC = twB + NMX_W64_TWB_N;
  const nmx_c2* tw2 = (const nmx_c2*)(tw_base + NMX_W64_TWL_FLOATS);   // [1024] w^k
  const unsigned tw2_addr = nmx_lds_addr(tw2 + l);
  nmx_c2 zlo[16], zhi[16], ve[16], vo[16];

  // ---- forward: x[4m .. 4m+3] -> (u[2m], u[2m+1]), two 1024-point transforms, radix-2 combine ----------------
  {
    const nmx_rsrc rs = nmx_make_rsrc(src, 4 * W);
    NMX_UNROLL
    for (int r = 0; r < 16; ++r) {
      const nmx_f4 q = __builtin_bit_cast(nmx_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * l + 1024 * r, 0, 0));
      ve[r] = nmx_mk2(q.x, q.y);
      vo[r] = nmx_mk2(q.z, q.w);
    }
    if (A.clean_on_load) {
      NMX_UNROLL
      for (int r = 0; r < 16; ++r) {
        ve[r] = nmx_mk2(nmx_clean_bl(ve[r].x), nmx_clean_bl(ve[r].y));
        vo[r] = nmx_mk2(nmx_clean_bl(vo[r].x), nmx_clean_bl(vo[r].y));
      }
    }
  }
  NMX_WSYNC();
  nmx_w64_fft1024<-1, 0>(ve, X, twB, twC, l);
  nmx_w64_fft1024<-1, 0>(vo, X, twB, twC, l);
  {
    nmx_c2 t2[16];   // w^k in NATURAL order: t2[r] = w^(l + 64 r); register i holds k = l + 64 NMX_I2J(i)
    nmx_ds_read_seq<512, 0>(t2, tw2_addr, std::make_integer_sequence<int, 16>{});
    nmx_lds_wait8(t2); nmx_lds_tie8(t2 + 8);
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) {
      const nmx_c2 wo = nmx_cmul_tw<0>(vo[i], t2[NMX_I2J(i)]);
      zlo[i] = nmx_cadd(ve[i], wo);
      zhi[i] = nmx_csub(ve[i], wo);
    }
  }

  const int nf = A.n_filters;
  for (int fi = 0; fi < nf; ++fi) {
    const NmxFilterDev& F = A.f[fi];
    // ---- spectral step + decimation-in-frequency split, four points of each half at a time ------------------
    const bool in_lds = fi < n_tab_lds;
    const unsigned ta = nmx_lds_addr(tab + (size_t)fi * 4096) + 8u * (unsigned)l;     // (A_k, B_k) pairs, k = l + 64 r
    const nmx_c2* tg = (const nmx_c2*)AA.Hs[fi] + l;                                   // the same table in global memory
    NMX_UNROLL
    for (int g = 0; g < 4; ++g) {
      // partners (cross-lane, compiler-tracked) first ...
      nmx_c2 pc[4], qc[4];
      NMX_UNROLL
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * g + q;
        const nmx_c2 zs = zhi[NMX_J2I(15 - r)], zt = zlo[NMX_J2I(15 - r)];
        pc[q] = nmx_mk2(__shfl(zs.x, (64 - l) & 63), __shfl(zs.y, (64 - l) & 63));   // partner of the LOW point k: high[1024 - k]
        qc[q] = nmx_mk2(__shfl(zt.x, (64 - l) & 63), __shfl(zt.y, (64 - l) & 63));   // partner of the HIGH point:  low[1024 - k]
        if (l == 0) {
          pc[q] = (r == 0) ? zlo[NMX_J2I(0)] : zhi[NMX_J2I((16 - r) & 15)];
          qc[q] = (r == 0) ? zhi[NMX_J2I(0)] : zlo[NMX_J2I((16 - r) & 15)];
        }
      }
      // ... then the table values: the unpaired LDS reads and their wait stay back to back (the compiler does
      // not know that the destination registers are in flight: nothing may move them in between)
      nmx_c2 tl[4], th[4], t2[4];
      if (in_lds) {
        nmx_ds_read_seq<512, 0>(tl, ta + 2048u * g, std::make_integer_sequence<int, 4>{});
        nmx_ds_read_seq<512, 8192>(th, ta + 2048u * g, std::make_integer_sequence<int, 4>{});
        nmx_ds_read_seq<512, 0>(t2, tw2_addr + 2048u * g, std::make_integer_sequence<int, 4>{});
        nmx_lds_wait5(tl[0], tl[1], tl[2], tl[3], th[0]);
        nmx_lds_tie2(th[1], th[2]); nmx_lds_tie2(th[3], t2[0]); nmx_lds_tie2(t2[1], t2[2]); nmx_lds_tie2(t2[3], pc[0]);
      } else {
        NMX_UNROLL
        for (int q = 0; q < 4; ++q) { tl[q] = tg[64 * (4 * g + q)]; th[q] = tg[64 * (4 * g + q) + 1024]; }
        nmx_ds_read_seq<512, 0>(t2, tw2_addr + 2048u * g, std::make_integer_sequence<int, 4>{});
        nmx_lds_wait5(t2[0], t2[1], t2[2], t2[3], pc[0]);
      }
      NMX_UNROLL
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * g + q;
        const nmx_c2 zl = nmx_axpby_swap_pair(tl[q], zlo[NMX_J2I(r)], pc[q]);
        const nmx_c2 zh = nmx_axpby_swap_pair(th[q], zhi[NMX_J2I(r)], qc[q]);
        ve[r] = nmx_cadd(zl, zh);
        vo[r] = nmx_cmul_tw<1>(nmx_csub(zl, zh), t2[q]);
      }
    }
    // ---- two inverse 1024-point transforms: ve -> x'[4m], x'[4m+1];  vo -> x'[4m+2], x'[4m+3] -----------------
    nmx_w64_fft1024<+1, HALF>(ve, X, twB, twC, l);
    nmx_w64_fft1024<+1, HALF>(vo, X, twB, twC, l);

    // ---- band-pass activity: tail variance, branch-free (as nmx_k_bank_w64p.h) ---------------------------------
    if (F.bp_seglen > 0) {
      const int lo = W - F.bp_seglen;
      const unsigned span = (unsigned)F.bp_seglen;
      const int s_l = 4 * l - lo;
      nmx_c2 acc = nmx_mk2(0.f, 0.f), acc2 = nmx_mk2(0.f, 0.f);
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        if (HALF && (i & 3) >= 2) continue;
        const int sb = s_l + 4 * (64 * (i >> 2) + 256 * (i & 3));
        nmx_c2 a = ve[i], b = vo[i];
        a.x = (unsigned)sb < span ? a.x : 0.f;
        a.y = (unsigned)(sb + 1) < span ? a.y : 0.f;
        b.x = (unsigned)(sb + 2) < span ? b.x : 0.f;
        b.y = (unsigned)(sb + 3) < span ? b.y : 0.f;
        acc = nmx_cadd(acc, nmx_cadd(a, b));
        acc2 = nmx_c2_fma(a, a, acc2);
        acc2 = nmx_c2_fma(b, b, acc2);
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
          const int sb = s_l + 4 * (64 * (i >> 2) + 256 * (i & 3));
          nmx_c2 a = nmx_csub(ve[i], mean2), b = nmx_csub(vo[i], mean2);
          a.x = (unsigned)sb < span ? a.x : 0.f;
          a.y = (unsigned)(sb + 1) < span ? a.y : 0.f;
          b.x = (unsigned)(sb + 2) < span ? b.x : 0.f;
          b.y = (unsigned)(sb + 3) < span ? b.y : 0.f;
          a2 = nmx_c2_fma(a, a, a2);
          a2 = nmx_c2_fma(b, b, a2);
        }
        tot = nmx_wave_reduce(a2.x + a2.y, 0.f, [](float a_, float b_) { return a_ + b_; });
      }
      const float act = tot / (float)F.bp_seglen;
      if (l == 0) {
        const int col = A.bp_cols.base + c * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
        out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
      }
    }
    // ---- filtered series to HBM: 16 bytes per lane and register, the ragged row end is the buffer range check ----
    float* dsw = F.sw_index >= 0 ? A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W : nullptr;
    float* dyb = F.burst_index >= 0 ? AA.yb_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W : nullptr;
    for (int dst = 0; dst < 2; ++dst) {
      float* d = dst ? dyb : dsw;
      if (!d) continue;
      const nmx_rsrc rs = nmx_make_rsrc(d, 4 * W);
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        if (HALF && (i & 3) >= 2) continue;
        const nmx_f4 q = {ve[i].x, ve[i].y, vo[i].x, vo[i].y};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nmx_u4, q), rs, 16 * l + 1024 * (i >> 2) + 4096 * (i & 3), 0, 0);
      }
    }
  }
}
#endif

// nmx_k_bank_w64c.h -- FIR bank, circular-convolution length M = 1536, ONE WAVE per (window, PAIR of channels).
//
// Reference arithmetic: filter/mne_filter.py:110-116 (zero-phase FIR, "same" part of the convolution of every
// window with every band's taps), features/bandpower.py:185-207 (variance of the band's tail segment).
//
// Two observations make this the cheapest exact formulation of the default shape (W = 1000, taps <= 999):
//   * The taps are symmetric, so their centred spectrum H is REAL, and only the W "same" samples are read.  A
//     circular convolution of length M aliases the discarded lead-in / lead-out onto each other, not onto the
//     window, as soon as M >= W + (L - 1) / 2 -- 1499 for the longest band, not W + L - 1 = 1998.  M = 1536
//     (= 64 lanes x 24 points) replaces the 2048-point convolution of nmx_k_bank_w64.h.
//   * Convolution with real taps is linear over C: conv(x1 + i x2, h) = conv(x1, h) + i conv(x2, h).  Two channels
//     of the same window ride in the real and imaginary part of ONE complex transform; the spectral step is the
//     pointwise product with the real H[k] (2 flops per point) -- the split / unsplit algebra of the half-length
//     real-FFT trick (conjugate partner from the mirrored lane, two tables, 4 flops per point) disappears.
//     Pairing is by channel, (2 i, 2 i + 1) of the SAME window: the result of an item does not depend on how the
//     hops are batched (batch == one-window call, bit for bit).  The second channel is scaled by a power of two
//     (exact) to the first one's magnitude before the transform and back after it, so a quiet channel next to a
//     loud one sees the same relative rounding noise as it would alone.
// Per item (one channel) this is 1536 log2(1536) / 2 butterfly work instead of 1024 log2(1024), half the exchange
// traffic, a twelfth of the table traffic: ~1 900 instead of ~3 400 VALU instructions.
//
// CDNA4 mapping: 24 complex points per lane, three register passes (radix 24 = 3 x 8, 8, 8) and two exchanges
// through one 13.5 KiB LDS tile per wave.  Forward = decimation in frequency (passes A, B, C), inverse = the
// mirror image (C', B', A'): the spectrum is consumed in the order the forward transform leaves it, and the series
// comes out in natural order (sample l + 64 j in register j of lane l: lane-consecutive stores, the W..M tail in
// registers that are never formed).  tools/model_w64c.py is the lane / register model of the index maps and
// proves every exchange access bank-conflict free per 32-lane half.
//   n = l + 64 j,  k = ka + 24 (qa + 8 qb),  ka = u + 8 g,  lane = l_lo + 8 u (passes A, B) or qa + 8 u (pass C)
//   P1: tile[ka(reg) 72 + l]                     A writes / A' reads      (reg 8 r + p holds ka = 3 p + r)
//   P2: tile[u 72 + l_lo + 576 g + 8 i]          B reads  / B' writes     (reg 8 g + i, i = l_hi)
//   P3: tile[u 72 + l_lo + 576 g + 9 i]          B writes / B' reads      (i = qa)
//   P4: tile[u 72 + 9 (lane & 7) + 576 g + i]    C reads  / C' writes     (i = l_lo)
// LDS per workgroup: H of every filter in register order (6 KiB each), exp(-2 pi i l ka / 1536) (12 KiB), one tile
// per wave -- 156 KiB for six filters and eight waves.  All tile and table accesses are volatile 8-byte LDS
// operations (unpaired; see nmx_device.h), in program order, so no fences are needed inside a wave.
// Device only; W <= 1024, activity-only band power, every filter with W + (L - 1) / 2 <= 1536.
#pragma once

#include "nmx_k_bank_w64.h"

#if !defined(NMX_HOST_EMU) && defined(NMX_LDS_ASM)

#define NMX_W64C_M 1536
#define NMX_W64C_TILE_FLOATS (2 * 24 * 72)        // one exchange tile (complex points: 24 rows of 64 + 8 pad)
#define NMX_W64C_TWA_FLOATS (2 * 24 * 64)         // exp(-2 pi i l ka(reg) / 1536), [reg][lane]
#define NMX_W64C_TWB_FLOATS (2 * 8 * 8)           // exp(-2 pi i a b / 64), [a][b]
#define NMX_W64C_H_FLOATS 1536                    // per filter: [12][64] pairs (H[k(lane, 2 i)], H[k(lane, 2 i + 1)])

template <int OFF>
NMX_DEV void nmx_ds_write_b64(unsigned addr, nmx_c2 v) {
  *(__attribute__((address_space(3))) volatile nmx_c2*)((__attribute__((address_space(3))) volatile char*)(unsigned long)addr + OFF) = v;
}
// a * (k, k), a * w and a * conj(w) with wave-uniform COMPILE-TIME constants: the pair sits in scalar registers
NMX_DEV nmx_c2 nmx_pk_mul_k(nmx_c2 a, float k) {
  const nmx_c2 kk = {k, k};
  nmx_c2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(kk));
  return r;
}
template <int CONJ>
NMX_DEV nmx_c2 nmx_cmul_k(nmx_c2 a, float wr, float wi) {
  const nmx_c2 w = {wr, wi};
  nmx_c2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  if (CONJ)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
// z * (h.x, h.x) and z * (h.y, h.y): the real H of two registers arrives as one 8-byte read
NMX_DEV nmx_c2 nmx_pk_mul_lo(nmx_c2 z, nmx_c2 h) {
  nmx_c2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(z), "v"(h));
  return r;
}
NMX_DEV nmx_c2 nmx_pk_mul_hi(nmx_c2 z, nmx_c2 h) {
  nmx_c2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(z), "v"(h));
  return r;
}

#define NMX_K_H8 0.70710678118654752f
#define NMX_K_S3 0.86602540378443865f

// a * exp(DIR 2 pi i MM / 24)
template <int DIR, int MM>
NMX_DEV nmx_c2 nmx_mul_w24(nmx_c2 a) {
  constexpr float C[15] = {1.f, 0.96592582628906829f, 0.86602540378443865f, 0.70710678118654752f, 0.5f,
                           0.25881904510252076f, 0.f, -0.25881904510252076f, -0.5f, -0.70710678118654752f,
                           -0.86602540378443865f, -0.96592582628906829f, -1.f, -0.96592582628906829f, -0.86602540378443865f};
  constexpr float S[15] = {0.f, 0.25881904510252076f, 0.5f, 0.70710678118654752f, 0.86602540378443865f,
                           0.96592582628906829f, 1.f, 0.96592582628906829f, 0.86602540378443865f, 0.70710678118654752f,
                           0.5f, 0.25881904510252076f, 0.f, -0.25881904510252076f, -0.5f};
  if constexpr (MM == 0) return a;
  else if constexpr (MM == 6) return nmx_mul_i<DIR>(a);
  else if constexpr (MM == 12) return -a;
  else if constexpr (MM == 3) return nmx_pk_mul_k(nmx_add_ib<DIR>(a, a), NMX_K_H8);   // h (1 + DIR i) a
  else return nmx_cmul_k<(DIR < 0)>(a, C[MM], S[MM]);
}

// 8-point DFT in place, natural order in and out, w = exp(DIR 2 pi i / 8): 28 packed operations
template <int DIR>
NMX_DEV void nmx_dft8(nmx_c2* a) {
  const nmx_c2 s0 = a[0] + a[4], d0 = a[0] - a[4];
  const nmx_c2 s1 = a[1] + a[5], d1 = a[1] - a[5];
  const nmx_c2 s2 = a[2] + a[6], d2 = a[2] - a[6];
  const nmx_c2 s3 = a[3] + a[7], d3 = a[3] - a[7];
  {   // even outputs: 4-point DFT of the sums
    const nmx_c2 t0 = s0 + s2, t1 = s0 - s2, t2 = s1 + s3, t3 = s1 - s3;
    a[0] = t0 + t2;
    a[4] = t0 - t2;
    a[2] = nmx_add_ib<DIR>(t1, t3);
    a[6] = nmx_add_ib<-DIR>(t1, t3);
  }
  {   // odd outputs: 4-point DFT of d_n w^n; the +-i and the sign of w^3 ride on the adds
    const nmx_c2 e1 = nmx_pk_mul_k(nmx_add_ib<DIR>(d1, d1), NMX_K_H8);    //  w^1 d1
    const nmx_c2 f3 = nmx_pk_mul_k(nmx_add_ib<-DIR>(d3, d3), NMX_K_H8);   // -w^3 d3
    const nmx_c2 t0 = nmx_add_ib<DIR>(d0, d2), t1 = nmx_add_ib<-DIR>(d0, d2);   // d0 +- w^2 d2
    const nmx_c2 t2 = e1 - f3, t3 = e1 + f3;
    a[1] = t0 + t2;
    a[5] = t0 - t2;
    a[3] = nmx_add_ib<DIR>(t1, t3);
    a[7] = nmx_add_ib<-DIR>(t1, t3);
  }
}

// ---- pass A (forward): 24-point DFT of v[j], j < 16 (the points j >= 16 are the zero padding) -------------------
// j = ja + 8 jb, ka = 3 p + r:  T[ja][r] = sum_jb v[ja + 8 jb] w3^(jb r);  * w24^(ja r);  DFT-8 over ja -> register 8 r + p
template <int JA>
NMX_DEV void nmx_dft24_fwd_col(nmx_c2* v) {
  const nmx_c2 a0 = v[JA], a1 = v[JA + 8];
  const nmx_c2 d = nmx_pk_mul_k(a1, NMX_K_S3);
  const nmx_c2 m = nmx_c2_fma(a1, nmx_mk2(-0.5f, -0.5f), a0);
  v[JA] = a0 + a1;
  v[JA + 8] = nmx_mul_w24<-1, JA>(nmx_add_ib<-1>(m, d));
  v[JA + 16] = nmx_mul_w24<-1, 2 * JA>(nmx_add_ib<+1>(m, d));
}
NMX_DEV void nmx_dft24_fwd(nmx_c2* v) {
  nmx_dft24_fwd_col<0>(v); nmx_dft24_fwd_col<1>(v); nmx_dft24_fwd_col<2>(v); nmx_dft24_fwd_col<3>(v);
  nmx_dft24_fwd_col<4>(v); nmx_dft24_fwd_col<5>(v); nmx_dft24_fwd_col<6>(v); nmx_dft24_fwd_col<7>(v);
  nmx_dft8<-1>(v);
  nmx_dft8<-1>(v + 8);
  nmx_dft8<-1>(v + 16);
}
// ---- pass A' (inverse): register 8 r + p holds ka = 3 p + r in; v[j], j < 16, out (j >= 16 never formed) -----------
template <int JA>
NMX_DEV void nmx_dft24_inv_col(nmx_c2* v) {
  const nmx_c2 a0 = v[JA], a1 = nmx_mul_w24<+1, JA>(v[JA + 8]), a2 = nmx_mul_w24<+1, 2 * JA>(v[JA + 16]);
  const nmx_c2 t = a1 + a2, d = nmx_pk_mul_k(a1 - a2, NMX_K_S3);
  const nmx_c2 m = nmx_c2_fma(t, nmx_mk2(-0.5f, -0.5f), a0);
  v[JA] = a0 + t;
  v[JA + 8] = nmx_add_ib<+1>(m, d);
}
NMX_DEV void nmx_dft24_inv(nmx_c2* v) {
  nmx_dft8<+1>(v);
  nmx_dft8<+1>(v + 8);
  nmx_dft8<+1>(v + 16);
  nmx_dft24_inv_col<0>(v); nmx_dft24_inv_col<1>(v); nmx_dft24_inv_col<2>(v); nmx_dft24_inv_col<3>(v);
  nmx_dft24_inv_col<4>(v); nmx_dft24_inv_col<5>(v); nmx_dft24_inv_col<6>(v); nmx_dft24_inv_col<7>(v);
}

// ---- exchange patterns: byte offsets of register I relative to the lane's base address --------------------------
constexpr int nmx_w64c_p1(int i) { return 8 * 72 * (3 * (i & 7) + (i >> 3)); }
constexpr int nmx_w64c_p2(int i) { return 8 * (576 * (i >> 3) + 8 * (i & 7)); }
constexpr int nmx_w64c_p3(int i) { return 8 * (576 * (i >> 3) + 9 * (i & 7)); }
constexpr int nmx_w64c_p4(int i) { return 8 * (576 * (i >> 3) + (i & 7)); }
template <int... I> NMX_DEV void nmx_w64c_rd1(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<nmx_w64c_p1(I)>(a)), ...); }
template <int... I> NMX_DEV void nmx_w64c_rd2(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<nmx_w64c_p2(I)>(a)), ...); }
template <int... I> NMX_DEV void nmx_w64c_rd3(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<nmx_w64c_p3(I)>(a)), ...); }
template <int... I> NMX_DEV void nmx_w64c_rd4(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<nmx_w64c_p4(I)>(a)), ...); }
template <int... I> NMX_DEV void nmx_w64c_wr1(const nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { (nmx_ds_write_b64<nmx_w64c_p1(I)>(a, v[I]), ...); }
template <int... I> NMX_DEV void nmx_w64c_wr2(const nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { (nmx_ds_write_b64<nmx_w64c_p2(I)>(a, v[I]), ...); }
template <int... I> NMX_DEV void nmx_w64c_wr3(const nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { (nmx_ds_write_b64<nmx_w64c_p3(I)>(a, v[I]), ...); }
template <int... I> NMX_DEV void nmx_w64c_wr4(const nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { (nmx_ds_write_b64<nmx_w64c_p4(I)>(a, v[I]), ...); }
// table rows: register I of lane l at I * 512 + 8 l bytes
template <int... I> NMX_DEV void nmx_w64c_rdt(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<512 * I>(a)), ...); }

struct NmxW64cLane {   // per-lane constants of a wave, set up once per kernel
  unsigned a1, a2, a4;    // LDS byte addresses of the lane's base in patterns P1, P2 (= P3), P4
  unsigned twa;           // LDS byte address of the lane's column of the pass-A twiddles
  nmx_c2 twb[8];          // exp(-2 pi i (lane & 7) q / 64), q = 0..7
};
#define NMX_W64C_SEQ24 std::make_integer_sequence<int, 24>{}

// forward transform: v[j] = x[l + 64 j] (j < 16) -> v[8 g + qb] = X[u + 8 g + 24 ((lane & 7) + 8 qb)]
NMX_DEV void nmx_w64c_forward(nmx_c2* v, const NmxW64cLane& Ln) {
  // The scheduler may not move anything across NMX_SCHED_FENCE(): every batch of exchange reads is issued as
  // one block (24 in flight, consumed group by group behind counted waits) instead of being sunk, one read at
  // a time, to its first use -- that serialises the LDS latency 24 times per pass.
  nmx_c2 w[24];
  nmx_w64c_rdt(w, Ln.twa, NMX_W64C_SEQ24);    // in flight during the butterflies
  NMX_SCHED_FENCE();
  nmx_dft24_fwd(v);
  NMX_UNROLL
  for (int i = 1; i < 24; ++i) v[i] = nmx_cmul_tw<0>(v[i], w[i]);
  nmx_w64c_wr1(v, Ln.a1, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  nmx_w64c_rd2(v, Ln.a2, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int g = 0; g < 3; ++g) {
    nmx_dft8<-1>(v + 8 * g);
    NMX_UNROLL
    for (int q = 1; q < 8; ++q) v[8 * g + q] = nmx_cmul_tw<0>(v[8 * g + q], Ln.twb[q]);
  }
  nmx_w64c_wr3(v, Ln.a2, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  nmx_w64c_rd4(v, Ln.a4, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int g = 0; g < 3; ++g) nmx_dft8<-1>(v + 8 * g);
}
// inverse transform (unnormalised), the mirror image: spectrum in the forward transform's output order in,
// v[j] = y[l + 64 j], j < 16, out
NMX_DEV void nmx_w64c_inverse(nmx_c2* v, const NmxW64cLane& Ln) {
  NMX_UNROLL
  for (int g = 0; g < 3; ++g) {
    nmx_dft8<+1>(v + 8 * g);
    NMX_UNROLL
    for (int q = 1; q < 8; ++q) v[8 * g + q] = nmx_cmul_tw<1>(v[8 * g + q], Ln.twb[q]);
  }
  nmx_w64c_wr4(v, Ln.a4, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  nmx_w64c_rd3(v, Ln.a2, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int g = 0; g < 3; ++g) nmx_dft8<+1>(v + 8 * g);
  nmx_w64c_wr2(v, Ln.a2, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  nmx_c2 w[24];
  nmx_w64c_rd1(v, Ln.a1, NMX_W64C_SEQ24);
  nmx_w64c_rdt(w, Ln.twa, NMX_W64C_SEQ24);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int i = 1; i < 24; ++i) v[i] = nmx_cmul_tw<1>(v[i], w[i]);
  nmx_dft24_inv(v);
}

NMX_DEV void nmx_w64c_lane_setup(NmxW64cLane& Ln, const float* tile, const float* twa_lds, const float* twb_glob, int l) {
  const unsigned base = nmx_lds_addr(tile);
  Ln.a1 = base + 8u * (unsigned)l;
  Ln.a2 = base + 8u * (unsigned)((l >> 3) * 72 + (l & 7));
  Ln.a4 = base + 8u * (unsigned)((l >> 3) * 72 + 9 * (l & 7));
  Ln.twa = nmx_lds_addr(twa_lds) + 8u * (unsigned)l;
  const nmx_c2* tb = (const nmx_c2*)twb_glob + (l & 7);
  NMX_UNROLL
  for (int q = 0; q < 8; ++q) Ln.twb[q] = tb[8 * q];
}

// ---- epilogue of one filter of a channel-pair item (shared with the M = 2048 pair kernel, nmx_k_bank_w64e.h):
// v[j] = (y_c, y_{c+1})[l + 64 j], j < 16.  Band-pass activity = variance of the tail [W - seglen, W) of both channels at
// once (re / im); filtered series to HBM as lane-consecutive 4-byte stores, one row per channel.
NMX_DEV void nmx_w64c_epilogue(const NmxBankW64Args& AA, const NmxFilterDev& F, const nmx_c2* v, int w, int c, int l, bool two,
                               float* out_row) {
  const NmxBankArgs& A = AA.b;
  const int W = A.W;
  // ---- band-pass activity: variance of the tail [W - seglen, W), both channels at once (re / im) ------------------
  if (F.bp_seglen > 0) {
    const unsigned span = (unsigned)F.bp_seglen;
    const int s_l = l - (W - F.bp_seglen);
    nmx_c2 acc = nmx_mk2(0.f, 0.f), acc2 = nmx_mk2(0.f, 0.f);
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) {
      const float mk = (unsigned)(s_l + 64 * j) < span ? 1.f : 0.f;
      const nmx_c2 val = v[j] * mk;
      acc = acc + val;
      acc2 = nmx_c2_fma(val, val, acc2);
    }
    const float inv_n = 1.f / (float)F.bp_seglen;
    float t1 = nmx_wave_reduce(acc.x, 0.f, [](float a_, float b_) { return a_ + b_; });
    float t2 = nmx_wave_reduce(acc.y, 0.f, [](float a_, float b_) { return a_ + b_; });
    const float q1 = nmx_wave_reduce(acc2.x, 0.f, [](float a_, float b_) { return a_ + b_; });
    const float q2 = nmx_wave_reduce(acc2.y, 0.f, [](float a_, float b_) { return a_ + b_; });
    const float mean1 = t1 * inv_n, mean2 = t2 * inv_n;
    t1 = q1 - mean1 * t1;   // sum (y - mean)^2 = sum y^2 - mean sum y
    t2 = q2 - mean2 * t2;
    if (mean1 * mean1 * (float)F.bp_seglen > 4.f * t1 || mean2 * mean2 * (float)F.bp_seglen > 4.f * t2) {
      // wave-uniform, rare (a short tail of a slow band is almost a constant): mean-shifted like np.var
      nmx_c2 a2 = nmx_mk2(0.f, 0.f);
      const nmx_c2 mm = nmx_mk2(mean1, mean2);
      NMX_UNROLL
      for (int j = 0; j < 16; ++j) {
        const float mk = (unsigned)(s_l + 64 * j) < span ? 1.f : 0.f;
        const nmx_c2 d = (v[j] - mm) * mk;
        a2 = nmx_c2_fma(d, d, a2);
      }
      t1 = nmx_wave_reduce(a2.x, 0.f, [](float a_, float b_) { return a_ + b_; });
      t2 = nmx_wave_reduce(a2.y, 0.f, [](float a_, float b_) { return a_ + b_; });
    }
    if (l < 2 && (l == 0 || two)) {
      const float act = (l == 0 ? t1 : t2) * inv_n;
      const int col = A.bp_cols.base + (c + l) * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
      out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
    }
  }
  // ---- filtered series to HBM: lane-consecutive 4-byte stores, one row per channel ---------------------------------
  float* dsw = F.sw_index >= 0 ? A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W : nullptr;
  float* dyb = F.burst_index >= 0 ? AA.yb_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W : nullptr;
  for (int dst = 0; dst < 2; ++dst) {
    float* d = dst ? dyb : dsw;
    if (!d) continue;
#ifdef NMX_DEBUG_NO_YB   // (bound experiment, tools/exp_fuse_bound.sh: the launcher nulls yb_out after a few launches --
    if (dst == 1 && !AA.yb_out) continue;   // the band series of the previous, identical step stay in place)
#endif
    const long long next = (long long)(dst ? A.n_burst_bands : A.n_sw_filters) * W;   // the same band of channel c + 1
    const nmx_rsrc s1 = nmx_make_rsrc(d, 4 * W);
    const nmx_rsrc s2 = nmx_make_rsrc(d + next, two ? 4 * W : 0);
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j].x), s1, 4 * l + 256 * j, 0, NMX_SERIES_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j].y), s2, 4 * l + 256 * j, 0, NMX_SERIES_STORE_AUX);
    }
  }
}

// one item: window w, channels c and c + 1 (c even; c + 1 == n_channels: the second half is zeros)
NMX_DEV void nmx_bank_w64c_item(const NmxBankW64Args& AA, int w, int c, const NmxW64cLane& Ln, const float* htab) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const NmxBankArgs& A = AA.b;
  const int W = A.W;
  const int l = (int)(threadIdx.x & 63);
  const bool two = c + 1 < A.n_channels;
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  nmx_c2 v[24], z[24];
  NMX_PROF_DECL

  // ---- load: sample l + 64 j of channel c -> re, of channel c + 1 -> im (the row end is the buffer range check) ----
  const nmx_rsrc r1 = nmx_make_rsrc(src, 4 * W);
  const nmx_rsrc r2 = nmx_make_rsrc(src + A.ch_stride, two ? 4 * W : 0);
  NMX_UNROLL
  for (int j = 0; j < 16; ++j) {
    v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 4 * l + 256 * j, 0, 0));
    v[j].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 4 * l + 256 * j, 0, 0));
  }
  if (A.clean_on_load) {
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) v[j] = nmx_mk2(nmx_clean_bl(v[j].x), nmx_clean_bl(v[j].y));
  }
  if (A.dcf) {   // the offset the stream was split from (nmx_engine_dc.inc), on the samples that exist (not on the zero padding)
    const nmx_c2 dd = nmx_mk2(A.dcf[c], two ? A.dcf[c + 1] : 0.f);
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) {
      if (64 * j + 63 < W) v[j] += dd;
      else if (l + 64 * j < W) v[j] += dd;
    }
  }
  // ---- the second channel at the first one's scale: an exact power of two -------------------------------------------
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
  // a channel that is identically zero (flat, zero-filled, or the missing partner of the last odd channel) must
  // come out EXACTLY zero, as it does alone (log10(0) -> -inf -> nan_to_num in the reference): its half of the
  // result would otherwise hold the partner's rounding noise
  const nmx_c2 unscale = nmx_mk2(m1 > 0.f ? 1.f : 0.f, m2 > 0.f ? __builtin_amdgcn_ldexpf(1.f, -e) : 0.f);

  NMX_PROF(0)
  nmx_w64c_forward(v, Ln);
  NMX_UNROLL
  for (int i = 0; i < 24; ++i) z[i] = v[i];
  NMX_PROF(1)

  const int nf = A.n_filters;
  const unsigned h_addr = nmx_lds_addr(htab) + 8u * (unsigned)l;
  for (int fi = 0; fi < nf; ++fi) {
    const NmxFilterDev& F = A.f[fi];
    {   // ---- spectral step: Z'[k] = H[k] Z[k], H real -------------------------------------------------------------
      nmx_c2 h[12];
      nmx_w64c_rdt(h, h_addr + (unsigned)fi * (NMX_W64C_H_FLOATS * 4u), std::make_integer_sequence<int, 12>{});
      NMX_SCHED_FENCE();
      NMX_UNROLL
      for (int i = 0; i < 12; ++i) {
        v[2 * i] = nmx_pk_mul_lo(z[2 * i], h[i]);
        v[2 * i + 1] = nmx_pk_mul_hi(z[2 * i + 1], h[i]);
      }
    }
    NMX_PROF(2)
    nmx_w64c_inverse(v, Ln);
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) v[j] = v[j] * unscale;
    NMX_PROF(3)

    nmx_w64c_epilogue(AA, F, v, w, c, l, two, out_row);
    NMX_PROF(5)
  }
#if defined(NMX_BANK_PROFILE)
  if (w == 5 && (c == 2 || c == 40) && l == 0)
    printf("bank c profile w=%d c=%d: load+scale %lld | forward %lld | spectral %lld | inverse %lld | variance %lld | stores %lld "
           "(cycles, whole item, %d filters)\n", w, c, pf_acc[0], pf_acc[1], pf_acc[2], pf_acc[3], pf_acc[4], pf_acc[5], nf);
#endif
}
#endif

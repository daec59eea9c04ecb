// nmx_k_bank_w64e.h -- FIR filtering with circular-convolution length M = 2048, ONE WAVE per (window, PAIR of channels):
// the 32-points-per-lane sibling of nmx_k_bank_w64c.h for everything that does not fit M = 1536.
//
//   PAD = 0  "same" FIR of a zero-padded window (filter/mne_filter.py:110-116) for filters with W + (L - 1) / 2 in
//            (1536, 2048]: the 1651-tap pre-filters of the sharp-wave analysis at the default settings
//            (features/sharpwaves.py:127-154,242-251).
//   PAD = 1  the notch: MNE's _overlap_add_filter(phase="zero", pad="reflect_limited") (filter/notch_filter.py:78-93):
//            odd reflection by (L - 1) / 2 samples (at most min(L, W) - 1 of them), linear FIR, crop to the window.
//
// Why a pair kernel.  Both used to run one channel per wave on the half-length real-FFT trick (1024-point complex
// transform, conjugate partner from the mirrored lane, two tables, split / unsplit algebra: nmx_k_bank_w64.h).  Taps are
// symmetric, so H is real, and a convolution with real taps is linear over C: channel c rides in the real part, channel
// c + 1 in the imaginary part of ONE 2048-point complex transform, and the spectral step is a scaling by H[k].  Per
// channel that is 2048 log2(2048) / 2 butterfly work instead of 1024 log2(1024) plus the split, half the exchange traffic
// and a table an eighth the size.
//
// The reflected signal is laid out CIRCULARLY: q[m] = x_ext[m] for m < W + h and q[M + j] = x_ext[j] for -h <= j < 0
// (h = (L - 1) / 2), so that y[n] = sum_j h_c[j] q[n - j] lands on n = 0 .. W - 1 -- register j of lane l holds
// sample l + 64 j on the way in AND on the way out, no offset.  M >= W + 2 h keeps the two flanks apart.
//
// CDNA4 mapping: as nmx_k_bank_w64c.h with four groups of eight registers instead of three -- radix 32 (= 4 x 8), 8, 8,
// decimation in frequency forward, the mirror image back, two exchanges through one 18 KiB tile per wave, the same
// conflict-free patterns (tools/model_w64e.py is the lane / register model):
//   n = l + 64 j,  k = ka + 32 (qa + 8 qb),  ka = u + 8 g;  pass-A register 8 r + p holds ka = 4 p + r.
// Pruning: PAD = 0 reads 16 of the 32 input registers (W <= 1024), both modes form 16 of the 32 output registers.
// Device only; W <= 1024, activity-only band power.
#pragma once

#include "nmx_k_bank_w64c.h"

#if !defined(NMX_HOST_EMU) && defined(NMX_LDS_ASM)

#define NMX_W64E_M 2048
#define NMX_W64E_TILE_FLOATS (2 * 32 * 72)        // one exchange tile (complex points: 32 rows of 64 + 8 pad)
#define NMX_W64E_TWA_FLOATS (2 * 32 * 64)         // exp(-2 pi i l ka(reg) / 2048), [reg][lane]
#define NMX_W64E_H_FLOATS 2048                    // per filter: [16][64] pairs (H[k(lane, 2 i)], H[k(lane, 2 i + 1)])

// a * exp(DIR 2 pi i MM / 32), MM = ja * r <= 21
template <int DIR, int MM>
NMX_DEV nmx_c2 nmx_mul_w32(nmx_c2 a) {
  constexpr float C[22] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                           0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f,
                           -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                           -0.92387953251128674f, -0.98078528040323043f, -1.f, -0.98078528040323043f, -0.92387953251128674f,
                           -0.83146961230254524f, -0.70710678118654752f, -0.55557023301960218f};
  constexpr float S[22] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                           0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f,
                           0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                           0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f, -0.38268343236508977f,
                           -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f};
  if constexpr (MM == 0) return a;
  else if constexpr (MM == 8) return nmx_mul_i<DIR>(a);
  else if constexpr (MM == 16) return -a;
  else if constexpr (MM == 4) return nmx_pk_mul_k(nmx_add_ib<DIR>(a, a), NMX_K_H8);        // h (1 + DIR i) a
  else if constexpr (MM == 12) return nmx_pk_mul_k(nmx_add_ib<-DIR>(a, a), -NMX_K_H8);     // h (-1 + DIR i) a
  else if constexpr (MM == 20) return nmx_pk_mul_k(nmx_add_ib<DIR>(a, a), -NMX_K_H8);      // -h (1 + DIR i) a
  else return nmx_cmul_k<(DIR < 0)>(a, C[MM], S[MM]);
}

// ---- pass A (forward): 32-point DFT of v[j].  j = ja + 8 jb, ka = 4 p + r:
// T[ja][r] = sum_jb v[ja + 8 jb] w4^(jb r);  * w32^(ja r);  DFT-8 over ja -> register 8 r + p
template <int JA, int FULL>
NMX_DEV void nmx_dft32_fwd_col(nmx_c2* v) {
  nmx_c2 a0 = v[JA], a1 = v[JA + 8], a2, a3;
  if (FULL) {
    a2 = v[JA + 16]; a3 = v[JA + 24];
    nmx_dft4<-1>(a0, a1, a2, a3);
  } else {   // v[j] = 0 for j >= 16 (the zero padding of a window of <= 1024 samples)
    const nmx_c2 s = a0 + a1, d = a0 - a1;
    a2 = d;
    a3 = nmx_add_ib<+1>(a0, a1);
    a1 = nmx_add_ib<-1>(a0, a1);
    a0 = s;
  }
  v[JA] = a0;
  v[JA + 8] = nmx_mul_w32<-1, JA>(a1);
  v[JA + 16] = nmx_mul_w32<-1, 2 * JA>(a2);
  v[JA + 24] = nmx_mul_w32<-1, 3 * JA>(a3);
}
template <int FULL>
NMX_DEV void nmx_dft32_fwd(nmx_c2* v) {
  nmx_dft32_fwd_col<0, FULL>(v); nmx_dft32_fwd_col<1, FULL>(v); nmx_dft32_fwd_col<2, FULL>(v); nmx_dft32_fwd_col<3, FULL>(v);
  nmx_dft32_fwd_col<4, FULL>(v); nmx_dft32_fwd_col<5, FULL>(v); nmx_dft32_fwd_col<6, FULL>(v); nmx_dft32_fwd_col<7, FULL>(v);
  nmx_dft8<-1>(v);
  nmx_dft8<-1>(v + 8);
  nmx_dft8<-1>(v + 16);
  nmx_dft8<-1>(v + 24);
}
// ---- pass A' (inverse): register 8 r + p holds ka = 4 p + r in; v[j], j < 16, out (j >= 16 never formed) -----------
template <int JA>
NMX_DEV void nmx_dft32_inv_col(nmx_c2* v) {
  const nmx_c2 a0 = v[JA], a1 = nmx_mul_w32<+1, JA>(v[JA + 8]), a2 = nmx_mul_w32<+1, 2 * JA>(v[JA + 16]),
               a3 = nmx_mul_w32<+1, 3 * JA>(v[JA + 24]);
  const nmx_c2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
  v[JA] = s02 + s13;                        // jb = 0
  v[JA + 8] = nmx_add_ib<+1>(d02, d13);     // jb = 1: a0 + i a1 - a2 - i a3
}
NMX_DEV void nmx_dft32_inv(nmx_c2* v) {
  nmx_dft8<+1>(v);
  nmx_dft8<+1>(v + 8);
  nmx_dft8<+1>(v + 16);
  nmx_dft8<+1>(v + 24);
  nmx_dft32_inv_col<0>(v); nmx_dft32_inv_col<1>(v); nmx_dft32_inv_col<2>(v); nmx_dft32_inv_col<3>(v);
  nmx_dft32_inv_col<4>(v); nmx_dft32_inv_col<5>(v); nmx_dft32_inv_col<6>(v); nmx_dft32_inv_col<7>(v);
}

// ---- exchange patterns (byte offsets of register I relative to the lane's base address): P2 .. P4 are those of the
// 1536-point kernel with a fourth group of rows; P1 follows this transform's pass-A register order
constexpr int nmx_w64e_p1(int i) { return 8 * 72 * (4 * (i & 7) + (i >> 3)); }
template <int... I> NMX_DEV void nmx_w64e_rd1(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<nmx_w64e_p1(I)>(a)), ...); }
template <int... I> NMX_DEV void nmx_w64e_wr1(const nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { (nmx_ds_write_b64<nmx_w64e_p1(I)>(a, v[I]), ...); }
// table rows, registers I0 .. I0 + 15 of lane l at I * 512 + 8 l bytes
template <int I0, int... I> NMX_DEV void nmx_w64e_rdt16(nmx_c2* v, unsigned a, std::integer_sequence<int, I...>) { ((v[I] = nmx_ds_read_b64<512 * (I0 + I)>(a)), ...); }
#define NMX_W64E_SEQ32 std::make_integer_sequence<int, 32>{}
#define NMX_W64E_SEQ16 std::make_integer_sequence<int, 16>{}

// forward transform: v[j] = x[l + 64 j] -> v[8 g + qb] = X[u + 8 g + 32 ((lane & 7) + 8 qb)]
template <int FULL>
NMX_DEV void nmx_w64e_forward(nmx_c2* v, const NmxW64cLane& Ln) {
  nmx_dft32_fwd<FULL>(v);
  {   // pass-A twiddles in two batches of sixteen (32 at once next to v[32] and the saved spectrum do not fit 256 VGPRs)
    nmx_c2 w[16];
    nmx_w64e_rdt16<0>(w, Ln.twa, NMX_W64E_SEQ16);
    NMX_SCHED_FENCE();
    NMX_UNROLL
    for (int i = 1; i < 16; ++i) v[i] = nmx_cmul_tw<0>(v[i], w[i]);
    nmx_w64e_rdt16<16>(w, Ln.twa, NMX_W64E_SEQ16);
    NMX_SCHED_FENCE();
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) v[16 + i] = nmx_cmul_tw<0>(v[16 + i], w[i]);
  }
  nmx_w64e_wr1(v, Ln.a1, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  nmx_w64c_rd2(v, Ln.a2, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int g = 0; g < 4; ++g) {
    nmx_dft8<-1>(v + 8 * g);
    NMX_UNROLL
    for (int q = 1; q < 8; ++q) v[8 * g + q] = nmx_cmul_tw<0>(v[8 * g + q], Ln.twb[q]);
  }
  nmx_w64c_wr3(v, Ln.a2, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  nmx_w64c_rd4(v, Ln.a4, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int g = 0; g < 4; ++g) nmx_dft8<-1>(v + 8 * g);
}
// inverse transform (unnormalised), the mirror image: v[j] = y[l + 64 j], j < 16, out
NMX_DEV void nmx_w64e_inverse(nmx_c2* v, const NmxW64cLane& Ln) {
  NMX_UNROLL
  for (int g = 0; g < 4; ++g) {
    nmx_dft8<+1>(v + 8 * g);
    NMX_UNROLL
    for (int q = 1; q < 8; ++q) v[8 * g + q] = nmx_cmul_tw<1>(v[8 * g + q], Ln.twb[q]);
  }
  nmx_w64c_wr4(v, Ln.a4, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  nmx_w64c_rd3(v, Ln.a2, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  NMX_UNROLL
  for (int g = 0; g < 4; ++g) nmx_dft8<+1>(v + 8 * g);
  nmx_w64c_wr2(v, Ln.a2, NMX_W64E_SEQ32);
  NMX_SCHED_FENCE();
  nmx_w64e_rd1(v, Ln.a1, NMX_W64E_SEQ32);
  {
    nmx_c2 w[16];
    nmx_w64e_rdt16<0>(w, Ln.twa, NMX_W64E_SEQ16);
    NMX_SCHED_FENCE();
    NMX_UNROLL
    for (int i = 1; i < 16; ++i) v[i] = nmx_cmul_tw<1>(v[i], w[i]);
    nmx_w64e_rdt16<16>(w, Ln.twa, NMX_W64E_SEQ16);
    NMX_SCHED_FENCE();
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) v[16 + i] = nmx_cmul_tw<1>(v[16 + i], w[i]);
  }
  nmx_dft32_inv(v);
}

// Where register j of lane l comes from in the notch's circular layout (header): sixteen words per lane, two 16-bit
// entries each (j = 2 i, 2 i + 1): byte offset of the source sample | form (0 as is, 1: 2 x[0] - x, 2: 2 x[W - 1] - x);
// 0xfffc = beyond the reflected signal or MNE's reflection limit n_edge: the range-checked load returns 0.
NMX_DEV void nmx_w64e_reflect_lane(const NmxBankArgs& A, int l, unsigned* rt) {
  const int W = A.W, h = A.pad_half, ne = A.n_edge;
  NMX_UNROLL
  for (int i = 0; i < 16; ++i) {
    unsigned word = 0;
    NMX_UNROLL
    for (int u = 0; u < 2; ++u) {
      const int m = l + 64 * (2 * i + u);
      const int j = m < W + h ? m : (m >= NMX_W64E_M - h ? m - NMX_W64E_M : (1 << 20));
      unsigned e = 0xfffcu;
      if (j < 0) { if (-j <= ne) e = (unsigned)(4 * -j) | 1u; }
      else if (j < W) e = (unsigned)(4 * j);
      else if (j < W + h) { const int rr = j - (W - 1); if (rr <= ne) e = (unsigned)(4 * (W - 1 - rr)) | 2u; }
      word |= e << (16 * u);
    }
    rt[i] = word;
  }
}

// The same for a shape known at compile time (WC samples, HC = (L - 1) / 2 <= the reflection limit; the default 1 kHz x
// 1 s window with the 999-tap notch: 1000, 499): which of the four regions -- window | right flank | gap | left flank --
// register J of a lane falls into is a constant for all but three registers, so the loads carry immediate offsets
// (flanks: descending in the lane, 4 (63 - l) + constant), `2 x[edge] - x` is one packed subtraction and only the
// registers that straddle a region boundary need a lane mask.  (The table form keeps ~100 loop-invariant lane masks
// in scalar registers, more than a wave has.)
template <int J, int WC, int HC, bool CLEAN>
NMX_DEV nmx_c2 nmx_w64e_reflect_reg(const nmx_rsrc r1, const nmx_rsrc r2, int l, nmx_c2 x0_2, nmx_c2 xl_2) {
  constexpr int m0 = 64 * J, M = NMX_W64E_M;
  auto ld = [&](int off) -> nmx_c2 {
    nmx_c2 x = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, off, 0, 0)),
                       __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, off, 0, 0)));
    if (CLEAN) x = nmx_mk2(nmx_clean_bl(x.x), nmx_clean_bl(x.y));
    return x;
  };
  nmx_c2 acc = nmx_mk2(0.f, 0.f);
  if constexpr (m0 < WC) acc = ld(4 * l + 4 * m0);   // the window itself (lanes beyond its end read 0: range check)
  if constexpr (m0 + 63 >= WC && m0 < WC + HC) {     // right flank: 2 x[W - 1] - x[2 (W - 1) - m]
    static_assert(2 * (WC - 1) - m0 - 63 >= 0, "reflection beyond the window");
    const nmx_c2 t = xl_2 - ld(4 * (63 - l) + 4 * (2 * (WC - 1) - m0 - 63));
    if constexpr (m0 >= WC && m0 + 63 < WC + HC) acc = t;
    else acc = (l >= WC - m0 && l < WC + HC - m0) ? t : acc;
  }
  if constexpr (m0 + 63 >= M - HC) {                 // left flank: 2 x[0] - x[M - m]
    const nmx_c2 t = x0_2 - ld(4 * (63 - l) + 4 * (M - m0 - 63));
    if constexpr (m0 >= M - HC) acc = t;
    else acc = (l >= M - HC - m0) ? t : acc;
  }
  return acc;
}
template <int WC, int HC, bool CLEAN, int... J>
NMX_DEV void nmx_w64e_reflect_fixed(nmx_c2* v, const nmx_rsrc r1, const nmx_rsrc r2, int l, nmx_c2 x0_2, nmx_c2 xl_2,
                                    std::integer_sequence<int, J...>) {
  ((v[J] = nmx_w64e_reflect_reg<J, WC, HC, CLEAN>(r1, r2, l, x0_2, xl_2)), ...);
}

// one item: window w, channels c and c + 1 (c even; c + 1 == n_channels: the second half is zeros)
// WC, HC != 0: the notch of that shape (nmx_w64e_reflect_fixed); 0: the lane's table `rt`
template <int PAD, int WC = 0, int HC = 0>
NMX_DEV void nmx_bank_w64e_item(const NmxBankW64Args& AA, int w, int c, const NmxW64cLane& Ln, const float* htab,
                                const unsigned* rt) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const NmxBankArgs& A = AA.b;
  const int W = A.W;
  const int l = (int)(threadIdx.x & 63);
  const bool two = c + 1 < A.n_channels;
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  nmx_c2 v[32];
  const nmx_rsrc r1 = nmx_make_rsrc(src, 4 * W);
  const nmx_rsrc r2 = nmx_make_rsrc(src + A.ch_stride, two ? 4 * W : 0);
  constexpr int NJ = PAD ? 32 : 16;

  if (PAD && WC) {
    nmx_c2 x0 = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 0, 0, 0)),
                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 0, 0, 0)));
    nmx_c2 xl = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 4 * (WC - 1), 0, 0)),
                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 4 * (WC - 1), 0, 0)));
    if (A.clean_on_load) {
      x0 = nmx_mk2(nmx_clean_bl(x0.x), nmx_clean_bl(x0.y));
      xl = nmx_mk2(nmx_clean_bl(xl.x), nmx_clean_bl(xl.y));
      nmx_w64e_reflect_fixed<WC, HC, true>(v, r1, r2, l, x0 + x0, xl + xl, NMX_W64E_SEQ32);
    } else {
      nmx_w64e_reflect_fixed<WC, HC, false>(v, r1, r2, l, x0 + x0, xl + xl, NMX_W64E_SEQ32);
    }
  } else if (PAD) {
    // ---- the odd-reflected window straight from global memory (L2): offset and form of every sample from the lane's table
    NMX_UNROLL
    for (int j = 0; j < 32; ++j) {
      const unsigned e = (rt[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      const int off = (int)(e & 0xfffcu);
      v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, off, 0, 0));
      v[j].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, off, 0, 0));
    }
    nmx_c2 x0 = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 0, 0, 0)),
                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 0, 0, 0)));
    nmx_c2 xl = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 4 * (W - 1), 0, 0)),
                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 4 * (W - 1), 0, 0)));
    if (A.clean_on_load) {
      x0 = nmx_mk2(nmx_clean_bl(x0.x), nmx_clean_bl(x0.y));
      xl = nmx_mk2(nmx_clean_bl(xl.x), nmx_clean_bl(xl.y));
      NMX_UNROLL
      for (int j = 0; j < 32; ++j) v[j] = nmx_mk2(nmx_clean_bl(v[j].x), nmx_clean_bl(v[j].y));
    }
    const nmx_c2 x0_2 = x0 + x0, xl_2 = xl + xl, zero = nmx_mk2(0.f, 0.f);
    NMX_UNROLL
    for (int j = 0; j < 32; ++j) {
      const unsigned f = (rt[j >> 1] >> (16 * (j & 1))) & 3u;
      const nmx_c2 base = f == 1u ? x0_2 : (f == 2u ? xl_2 : zero);
      const float sg = f ? -1.f : 1.f;
      v[j] = nmx_c2_fma(v[j], nmx_mk2(sg, sg), base);
    }
  } else {
    // ---- load: sample l + 64 j of channel c -> re, of channel c + 1 -> im (the row end is the buffer range check) ----
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
  }
  // the notch in residual form (NmxBankArgs::residual) stores x - g * x_ext: the window's samples stay in registers
  // (the workgroup's exchange tiles bound the kernel at two waves per SIMD: 256 VGPRs each, 158 in use without them;
  // re-reading them from L2 behind the inverse transform cost 0.15 ms of the 0.71 ms launch)
  // (the table-driven form of the reflection is at 236 VGPRs already: it reads them again)
  constexpr bool KEEP = PAD && WC;
  nmx_c2 keep[KEEP ? 16 : 1];
  if constexpr (KEEP) {
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) keep[j] = v[j];
  }
  // ---- the second channel at the first one's scale: an exact power of two (nmx_k_bank_w64c.h) ------------------------
  float m1 = 0.f, m2 = 0.f;
  NMX_UNROLL
  for (int j = 0; j < NJ; ++j) { m1 = fmaxf(m1, fabsf(v[j].x)); m2 = fmaxf(m2, fabsf(v[j].y)); }
  m1 = nmx_wave_reduce(m1, 0.f, [](float a_, float b_) { return fmaxf(a_, b_); });
  m2 = nmx_wave_reduce(m2, 0.f, [](float a_, float b_) { return fmaxf(a_, b_); });
  int e = 0;
  if (m1 > 0.f && m2 > 0.f) {
    e = __builtin_amdgcn_frexp_expf(m1) - __builtin_amdgcn_frexp_expf(m2);
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
  }
  NMX_UNROLL
  for (int j = 0; j < NJ; ++j) v[j].y = __builtin_amdgcn_ldexpf(v[j].y, e);
  // (a channel that is identically zero comes out EXACTLY zero, as it does alone)
  const nmx_c2 unscale = nmx_mk2(m1 > 0.f ? 1.f : 0.f, m2 > 0.f ? __builtin_amdgcn_ldexpf(1.f, -e) : 0.f);

  nmx_w64e_forward<PAD>(v, Ln);
  const unsigned h_addr = nmx_lds_addr(htab) + 8u * (unsigned)l;

  if (PAD) {
    // ---- ONE filter: scale the spectrum in place, transform back, store the window ------------------------------------
    {
      nmx_c2 h[16];
      nmx_w64e_rdt16<0>(h, h_addr, NMX_W64E_SEQ16);
      NMX_SCHED_FENCE();
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        v[2 * i] = nmx_pk_mul_lo(v[2 * i], h[i]);
        v[2 * i + 1] = nmx_pk_mul_hi(v[2 * i + 1], h[i]);
      }
    }
    nmx_w64e_inverse(v, Ln);
    float* d = A.y_out + ((long long)w * A.n_channels + c) * W;
    const nmx_rsrc s1 = nmx_make_rsrc(d, 4 * W);
    const nmx_rsrc s2 = nmx_make_rsrc(d + W, two ? 4 * W : 0);
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) {
      nmx_c2 y = v[j] * unscale;
      if (A.residual) {
        if constexpr (KEEP) {
          y = keep[j] - y;   // (lanes beyond the window's end hold flank samples: their stores are out of range)
        } else {
          nmx_c2 xw = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, 4 * l + 256 * j, 0, 0)),
                              __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, 4 * l + 256 * j, 0, 0)));
          if (A.clean_on_load) xw = nmx_mk2(nmx_clean_bl(xw.x), nmx_clean_bl(xw.y));
          y = xw - y;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y.x), s1, 4 * l + 256 * j, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y.y), s2, 4 * l + 256 * j, 0, 0);
    }
    return;
  }

  nmx_c2 z[32];
  NMX_UNROLL
  for (int i = 0; i < 32; ++i) z[i] = v[i];
  const int nf = A.n_filters;
  for (int fi = 0; fi < nf; ++fi) {
    const NmxFilterDev& F = A.f[fi];
    {   // ---- spectral step: Z'[k] = H[k] Z[k], H real -------------------------------------------------------------
      nmx_c2 h[16];
      nmx_w64e_rdt16<0>(h, h_addr + (unsigned)fi * (NMX_W64E_H_FLOATS * 4u), NMX_W64E_SEQ16);
      NMX_SCHED_FENCE();
      NMX_UNROLL
      for (int i = 0; i < 16; ++i) {
        v[2 * i] = nmx_pk_mul_lo(z[2 * i], h[i]);
        v[2 * i + 1] = nmx_pk_mul_hi(z[2 * i + 1], h[i]);
      }
    }
    nmx_w64e_inverse(v, Ln);
    NMX_UNROLL
    for (int j = 0; j < 16; ++j) v[j] = v[j] * unscale;
    nmx_w64c_epilogue(AA, F, v, w, c, l, two, out_row);
  }
}
#endif

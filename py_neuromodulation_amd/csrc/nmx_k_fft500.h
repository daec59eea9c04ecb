// nmx_k_fft500.h -- wave-level complex transform of length 500 (= real length 1000, the default
// window) and the Hilbert-envelope kernel built on it.
//
// The LDS Stockham transforms of nmx_device.h run one butterfly per thread with the twiddle powers
// derived per butterfly and a workgroup barrier per stage; in the radix-10 stages 50 of 128 threads
// work.  Here ONE wave owns the transform (device only):
//   * stages 10 . 10 . 5 (Stockham autosort, same index maps as nmx_stage_static), lanes 0..49 hold one
//     radix-10 butterfly (10 points in VGPRs) in stages 1 and 2 and two radix-5 butterflies in stage 3;
//   * all 17 twiddles a lane ever needs are loaded ONCE from the plan's table into registers and
//     serve the forward and the inverse transform (conjugation is a sign modifier);
//   * every index is  lane + compile-time constant,  stages are separated by wave-local LDS fences.
#pragma once

#include "nmx_device.h"

template <int DIR>
NMX_DEV nmx_c2 nmx_twd(nmx_c2 t) { return DIR > 0 ? nmx_mk2(t.x, -t.y) : t; }

// Tables of the wave-level transform, built once on the host (nmx_engine.inc: build_hilbert), floats:
//   tw[i * 64 + lane], i = 0..16 (complex): the 17 twiddles lane `lane` needs --
//     i = 0..8   stage 2: exp(-2 pi i 5 k r / 500), k = lane % 10, r = i + 1
//     i = 9..12  stage 3, butterfly j = lane:      exp(-2 pi i j r / 500), r = i - 8
//     i = 13..16 stage 3, butterfly j = lane + 50: r = i - 12          (lanes >= 50: copies of lane 0)
//   cs[k], k = 0..499 (complex): (2 cos(2 pi k / 1000), 2 sin(2 pi k / 1000)) / 1000, cs[0] = 0 --
//     the spectral step of the Hilbert transform (nmx_w500_hilbert)
#define NMX_W500_TW_N (17 * 64)
#define NMX_W500_CS_N 500
#define NMX_W500_TAB_FLOATS (2 * (NMX_W500_TW_N + NMX_W500_CS_N))

#ifndef NMX_HOST_EMU

// twiddle providers: registers (loaded once per wave from the global table) or an LDS copy
struct NmxW500TwReg {
  nmx_c2 a[17];
  NMX_DEV void load(const float* tab, int lane) {
#pragma unroll
    for (int i = 0; i < 17; ++i) a[i] = ((const nmx_c2*)tab)[i * 64 + lane];
  }
  NMX_DEV nmx_c2 get(int i) const { return a[i]; }
};
// 11 instead of 17 twiddle registers: the stage-3 twiddles w^(j r), r = 2, 3, 4, are formed from w^j when they are
// asked for (three complex products per half) -- for kernels that are short of registers, not of issue slots
struct NmxW500TwRegC {
  nmx_c2 a[9], b[2];
  NMX_DEV void load(const float* tab, int lane) {
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = ((const nmx_c2*)tab)[i * 64 + lane];
    b[0] = ((const nmx_c2*)tab)[9 * 64 + lane];
    b[1] = ((const nmx_c2*)tab)[13 * 64 + lane];
  }
  NMX_DEV nmx_c2 get(int i) const {
    if (i < 9) return a[i];
    const nmx_c2 w1 = b[(i - 9) >> 2];
    const int r = (i - 9) & 3;   // 0..3: w^1 .. w^4
    if (r == 0) return w1;
    const nmx_c2 w2 = nmx_cmul(w1, w1);
    if (r == 1) return w2;
    if (r == 2) return nmx_cmul(w2, w1);
    return nmx_cmul(w2, w2);
  }
};
struct NmxW500TwLds {
  const nmx_c2* p;   // table + lane
  NMX_DEV nmx_c2 get(int i) const { return p[64 * i]; }
};

template <int DIR>
NMX_DEV void nmx_dft5_c2(nmx_c2& x0, nmx_c2& x1, nmx_c2& x2, nmx_c2& x3, nmx_c2& x4) {
  const float c1 = 0.30901699437494745f, c2 = -0.80901699437494745f;
  const float s1 = DIR * 0.95105651629515353f, s2 = DIR * 0.58778525229247314f;
  const nmx_c2 t1 = x1 + x4, t2 = x2 + x3, d1 = x1 - x4, d2 = x2 - x3;
  const nmx_c2 m1 = x0 + c1 * t1 + c2 * t2, m2 = x0 + c2 * t1 + c1 * t2;
  const nmx_c2 u1 = s1 * d1 + s2 * d2, u2 = s2 * d1 - s1 * d2;
  x0 = x0 + t1 + t2;
  // m +- i u: the +-i rides on the add (nmx_add_ib: op_sel / neg modifiers, one instruction)
  x1 = nmx_add_ib<+1>(m1, u1); x4 = nmx_add_ib<-1>(m1, u1);
  x2 = nmx_add_ib<+1>(m2, u2); x3 = nmx_add_ib<-1>(m2, u2);
}

// 10-point DFT of v[0..9] in place (output index q in natural order), 10 = 2 x 5
template <int DIR>
NMX_DEV void nmx_dft10_c2(nmx_c2* v) {
  nmx_c2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], e4 = v[8];
  nmx_c2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7], o4 = v[9];
  nmx_dft5_c2<DIR>(e0, e1, e2, e3, e4);
  nmx_dft5_c2<DIR>(o0, o1, o2, o3, o4);
  const float sg = (float)DIR;
  o1 = nmx_cmul(o1, nmx_mk2(0.80901699437494745f, sg * 0.58778525229247314f));
  o2 = nmx_cmul(o2, nmx_mk2(0.30901699437494745f, sg * 0.95105651629515353f));
  o3 = nmx_cmul(o3, nmx_mk2(-0.30901699437494745f, sg * 0.95105651629515353f));
  o4 = nmx_cmul(o4, nmx_mk2(-0.80901699437494745f, sg * 0.58778525229247314f));
  v[0] = e0 + o0; v[1] = e1 + o1; v[2] = e2 + o2; v[3] = e3 + o3; v[4] = e4 + o4;
  v[5] = e0 - o0; v[6] = e1 - o1; v[7] = e2 - o2; v[8] = e3 - o3; v[9] = e4 - o4;
}

// in -> a -> b -> a ; returns a (natural order).  `in` may alias b.  All three are wave-private.
template <int DIR, typename TW>
NMX_DEV nmx_c2* nmx_w500_fft(const nmx_c2* in, nmx_c2* a, nmx_c2* b, const TW& T, int lane) {
  nmx_c2 v[10];
  if (lane < 50) {
    // stage 1: R = 10, Ns = 1: in[j + 50 r] -> a[10 j + r]   (unpaired ds_read_b64: nmx_device.h)
    nmx_ds_read_seq<400, 0>(v, nmx_lds_addr(in + lane), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]);
    nmx_dft10_c2<DIR>(v);
    nmx_c2* o = a + 10 * lane;
#pragma unroll
    for (int r = 0; r < 10; ++r) o[r] = v[r];
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    // stage 2: R = 10, Ns = 10: a[j + 50 r] * w^(5 k r) -> b[100 q + k + 10 r],  q = j / 10, k = j % 10
    nmx_ds_read_seq<400, 0>(v, nmx_lds_addr(a + lane), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]);
#pragma unroll
    for (int r = 1; r < 10; ++r) v[r] = nmx_cmul_tw<(DIR > 0)>(v[r], T.get(r - 1));
    nmx_dft10_c2<DIR>(v);
    const int q = lane / 10, k = lane - 10 * q;
    nmx_c2* o = b + 100 * q + k;
#pragma unroll
    for (int r = 0; r < 10; ++r) o[10 * r] = v[r];
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    // stage 3: R = 5, Ns = 100: b[j + 100 r] * w^(j r) -> a[j + 100 r],  j = lane and lane + 50
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      nmx_c2 x[5];
      nmx_ds_read_seq<800, 0>(x, nmx_lds_addr(b + lane + 50 * h), std::make_integer_sequence<int, 5>{});
      nmx_lds_wait5(x[0], x[1], x[2], x[3], x[4]);
      nmx_c2 x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4];
      x1 = nmx_cmul_tw<(DIR > 0)>(x1, T.get(9 + 4 * h));
      x2 = nmx_cmul_tw<(DIR > 0)>(x2, T.get(10 + 4 * h));
      x3 = nmx_cmul_tw<(DIR > 0)>(x3, T.get(11 + 4 * h));
      x4 = nmx_cmul_tw<(DIR > 0)>(x4, T.get(12 + 4 * h));
      nmx_dft5_c2<DIR>(x0, x1, x2, x3, x4);
      nmx_c2* o = a + lane + 50 * h;
      o[0] = x0; o[100] = x1; o[200] = x2; o[300] = x3; o[400] = x4;
    }
  }
  NMX_WAVE_FENCE();
  return a;
}

// Forward transform when only the bins k < k_hi and k > 500 - k_hi are read afterwards (band powers up to a
// few tens of Hz: k_hi <= 100).  Stages 1 and 2 as above; in stage 3 (radix 5, outputs a[j + 100 r]) only the
// butterflies j < k_hi (their r = 0 output) and j > 100 - k_hi (their r = 4 output) run, and each forms that
// ONE output -- with the same operations the full butterfly uses for it, so the values are identical.
template <typename TW>
NMX_DEV nmx_c2* nmx_w500_fft_fwd_low(const nmx_c2* in, nmx_c2* a, nmx_c2* b, const TW& T, int lane, int k_hi) {
  nmx_c2 v[10];
  if (lane < 50) {
    nmx_ds_read_seq<400, 0>(v, nmx_lds_addr(in + lane), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]);
    nmx_dft10_c2<-1>(v);
    nmx_c2* o = a + 10 * lane;
#pragma unroll
    for (int r = 0; r < 10; ++r) o[r] = v[r];
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    nmx_ds_read_seq<400, 0>(v, nmx_lds_addr(a + lane), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]);
#pragma unroll
    for (int r = 1; r < 10; ++r) v[r] = nmx_cmul_tw<0>(v[r], T.get(r - 1));
    nmx_dft10_c2<-1>(v);
    const int q = lane / 10, k = lane - 10 * q;
    nmx_c2* o = b + 100 * q + k;
#pragma unroll
    for (int r = 0; r < 10; ++r) o[10 * r] = v[r];
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    const float c1 = 0.30901699437494745f, c2 = -0.80901699437494745f;
    const float s1 = -0.95105651629515353f, s2 = -0.58778525229247314f;   // DIR = -1
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lane + 50 * h;
      const bool lo = j < k_hi, hi = j > 100 - k_hi;
      if (!(lo || hi)) continue;
      const nmx_c2* src = b + j;
      nmx_c2 x0 = src[0], x1 = src[100], x2 = src[200], x3 = src[300], x4 = src[400];
      x1 = nmx_cmul_tw<0>(x1, T.get(9 + 4 * h));
      x2 = nmx_cmul_tw<0>(x2, T.get(10 + 4 * h));
      x3 = nmx_cmul_tw<0>(x3, T.get(11 + 4 * h));
      x4 = nmx_cmul_tw<0>(x4, T.get(12 + 4 * h));
      const nmx_c2 t1 = x1 + x4, t2 = x2 + x3;
      if (lo) a[j] = x0 + t1 + t2;
      if (hi) {
        const nmx_c2 d1 = x1 - x4, d2 = x2 - x3;
        const nmx_c2 m1 = x0 + c1 * t1 + c2 * t2;
        const nmx_c2 u1 = s1 * d1 + s2 * d2;
        a[j + 400] = nmx_add_ib<-1>(m1, u1);
      }
    }
  }
  NMX_WAVE_FENCE();
  return a;
}

// Hilbert transform H[y] of a real series of 1000 samples (scipy.signal.hilbert's imaginary part).
// In: b[m] = (y[2m], y[2m+1]), m < 500.  Out (returned pointer, = a): (H[y][2m], H[y][2m+1]).
// With Z = FFT_500(b), th = 2 pi k / 1000, the half-length forward split, the multiplication by -i
// (DC and Nyquist dropped) and the half-length inverse unsplit collapse to
//     Z'[k] = 2 (cos(th) conj(Z[500 - k]) + i sin(th) Z[k]),   Z'[0] = 0,
// and H[y] = IFFT_500(Z') / 1000 -- four flops per point with one table (cs, normalisation folded in).
template <typename TW>
NMX_DEV const nmx_c2* nmx_w500_hilbert(nmx_c2* a, nmx_c2* b, const TW& T, const nmx_c2* cs, int lane) {
  const nmx_c2* Z = nmx_w500_fft<-1>(b, a, b, T, lane);   // = a
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = lane + 64 * q;
    if (k < 500) {
      const nmx_c2 zk = Z[k], zc = Z[k == 0 ? 0 : 500 - k], w = cs[k];
      b[k] = nmx_mk2(w.x * zc.x - w.y * zk.y, w.y * zk.x - w.x * zc.y);
    }
  }
  NMX_WAVE_FENCE();
  return nmx_w500_fft<+1>(b, a, b, T, lane);
}

// ---- 1000 points (real length 2000: BASELINE config 3, 2 kHz x 1 s windows) ------------------------------------------
// Stages 10 . 10 . 10 (the same Stockham index maps), 100 radix-10 butterflies per stage: lane l < 50 takes butterflies
// l and l + 50 -- they share the stage-2 twiddles ((l + 50) % 10 = l % 10).  27 twiddles per lane:
//   tw[i * 64 + lane], i = 0..8    stage 2: exp(-2 pi i 10 k r / 1000), k = lane % 10, r = i + 1
//                      i = 9..17   stage 3, butterfly j = lane:      exp(-2 pi i j r / 1000), r = i - 8
//                      i = 18..26  stage 3, butterfly j = lane + 50: r = i - 17          (lanes >= 50: copies of lane 0)
//   cs[k], k = 0..999 (complex): (2 cos(2 pi k / 2000), 2 sin(2 pi k / 2000)) / 2000, cs[0] = 0
#define NMX_W1000_TW_N (27 * 64)
#define NMX_W1000_CS_N 1000
#define NMX_W1000_TAB_FLOATS (2 * (NMX_W1000_TW_N + NMX_W1000_CS_N))
struct NmxW1000TwReg {
  nmx_c2 a[27];
  NMX_DEV void load(const float* tab, int lane) {
#pragma unroll
    for (int i = 0; i < 27; ++i) a[i] = ((const nmx_c2*)tab)[i * 64 + lane];
  }
  NMX_DEV nmx_c2 get(int i) const { return a[i]; }
};

// in -> a -> b -> a ; returns a (natural order).  A lane reads the points of BOTH its butterflies before it writes any, and
// a wave executes in lockstep, so every stage may run IN PLACE: in, a and b may all be the same 1000-point buffer (8 KB per
// wave instead of 16: twice the resident waves).
template <int DIR, typename TW>
NMX_DEV nmx_c2* nmx_w1000_fft(const nmx_c2* in, nmx_c2* a, nmx_c2* b, const TW& T, int lane) {
  nmx_c2 v[10], u[10];
  if (lane < 50) {
    // stage 1: R = 10, Ns = 1: in[j + 100 r] -> a[10 j + r]
    nmx_ds_read_seq<800, 0>(v, nmx_lds_addr(in + lane), std::make_integer_sequence<int, 10>{});
    nmx_ds_read_seq<800, 0>(u, nmx_lds_addr(in + lane + 50), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]); nmx_lds_tie8(u); nmx_lds_tie2(u[8], u[9]);
    nmx_dft10_c2<DIR>(v);
    nmx_dft10_c2<DIR>(u);
  }
  NMX_WAVE_FENCE();   // (every lane's reads are done: in-place stores are safe)
  if (lane < 50) {
    nmx_c2* o = a + 10 * lane;
#pragma unroll
    for (int r = 0; r < 10; ++r) { o[r] = v[r]; o[500 + r] = u[r]; }
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    // stage 2: R = 10, Ns = 10: a[j + 100 r] * w^(10 k r) -> b[100 q + k + 10 r],  q = j / 10, k = j % 10
    nmx_ds_read_seq<800, 0>(v, nmx_lds_addr(a + lane), std::make_integer_sequence<int, 10>{});
    nmx_ds_read_seq<800, 0>(u, nmx_lds_addr(a + lane + 50), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]); nmx_lds_tie8(u); nmx_lds_tie2(u[8], u[9]);
#pragma unroll
    for (int r = 1; r < 10; ++r) { v[r] = nmx_cmul_tw<(DIR > 0)>(v[r], T.get(r - 1)); u[r] = nmx_cmul_tw<(DIR > 0)>(u[r], T.get(r - 1)); }
    nmx_dft10_c2<DIR>(v);
    nmx_dft10_c2<DIR>(u);
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    const int q0 = lane / 10, k = lane - 10 * q0;
    nmx_c2* o = b + 100 * q0 + k;
#pragma unroll
    for (int r = 0; r < 10; ++r) { o[10 * r] = v[r]; o[500 + 10 * r] = u[r]; }
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    // stage 3: R = 10, Ns = 100: b[j + 100 r] * w^(j r) -> a[j + 100 r]
    nmx_ds_read_seq<800, 0>(v, nmx_lds_addr(b + lane), std::make_integer_sequence<int, 10>{});
    nmx_ds_read_seq<800, 0>(u, nmx_lds_addr(b + lane + 50), std::make_integer_sequence<int, 10>{});
    nmx_lds_wait8(v); nmx_lds_tie2(v[8], v[9]); nmx_lds_tie8(u); nmx_lds_tie2(u[8], u[9]);
#pragma unroll
    for (int r = 1; r < 10; ++r) { v[r] = nmx_cmul_tw<(DIR > 0)>(v[r], T.get(9 + r - 1)); u[r] = nmx_cmul_tw<(DIR > 0)>(u[r], T.get(18 + r - 1)); }
    nmx_dft10_c2<DIR>(v);
    nmx_dft10_c2<DIR>(u);
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    nmx_c2* o = a + lane;
#pragma unroll
    for (int r = 0; r < 10; ++r) { o[100 * r] = v[r]; o[50 + 100 * r] = u[r]; }
  }
  NMX_WAVE_FENCE();
  return a;
}

// Hilbert transform of a real series of 2000 samples: nmx_w500_hilbert with n = 1000 (same collapsed spectral step)
template <typename TW>
NMX_DEV const nmx_c2* nmx_w1000_hilbert(nmx_c2* a, nmx_c2* b, const TW& T, const nmx_c2* cs, int lane) {
  const nmx_c2* Z = nmx_w1000_fft<-1>(b, a, b, T, lane);   // = a
  nmx_c2 zp[16];   // (a may be b: the whole spectrum is read before any point of Z' is written)
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = lane + 64 * q;
    zp[q] = nmx_mk2(0.f, 0.f);
    if (k < 1000) {
      const nmx_c2 zk = Z[k], zc = Z[k == 0 ? 0 : 1000 - k], w = cs[k];
      zp[q] = nmx_mk2(w.x * zc.x - w.y * zk.y, w.y * zk.x - w.x * zc.y);
    }
  }
  NMX_WAVE_FENCE();
#pragma unroll
  for (int q = 0; q < 16; ++q)
    if (lane + 64 * q < 1000) b[lane + 64 * q] = zp[q];
  NMX_WAVE_FENCE();
  return nmx_w1000_fft<+1>(b, a, b, T, lane);
}
#endif

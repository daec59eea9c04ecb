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

#include "nmx_k_bank_w64.h"

#ifndef NMX_HOST_EMU

struct NmxW500Tw {
  nmx_c2 s2[9];    // stage 2: exp(-2 pi i 5 k r / 500), k = lane % 10, r = 1..9
  nmx_c2 s3a[4];   // stage 3, butterfly j = lane:      exp(-2 pi i j r / 500), r = 1..4
  nmx_c2 s3b[4];   // stage 3, butterfly j = lane + 50
};

NMX_DEV void nmx_w500_load_tw(NmxW500Tw& T, const float2* tw, int lane) {
  const int l = lane < 50 ? lane : 0, k = l % 10;
#pragma unroll
  for (int r = 1; r < 10; ++r) T.s2[r - 1] = nmx_to_c2(tw[(5 * k * r) % 500]);
#pragma unroll
  for (int r = 1; r < 5; ++r) {
    T.s3a[r - 1] = nmx_to_c2(tw[(l * r) % 500]);
    T.s3b[r - 1] = nmx_to_c2(tw[((l + 50) * r) % 500]);
  }
}

template <int DIR>
NMX_DEV void nmx_dft5_c2(nmx_c2& x0, nmx_c2& x1, nmx_c2& x2, nmx_c2& x3, nmx_c2& x4) {
  const float c1 = 0.30901699437494745f, c2 = -0.80901699437494745f;
  const float s1 = DIR * 0.95105651629515353f, s2 = DIR * 0.58778525229247314f;
  const nmx_c2 t1 = x1 + x4, t2 = x2 + x3, d1 = x1 - x4, d2 = x2 - x3;
  const nmx_c2 m1 = x0 + c1 * t1 + c2 * t2, m2 = x0 + c2 * t1 + c1 * t2;
  const nmx_c2 u1 = s1 * d1 + s2 * d2, u2 = s2 * d1 - s1 * d2;
  const nmx_c2 n1 = {-u1.y, u1.x}, n2 = {-u2.y, u2.x};
  x0 = x0 + t1 + t2;
  x1 = m1 + n1; x4 = m1 - n1;
  x2 = m2 + n2; x3 = m2 - n2;
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
template <int DIR>
NMX_DEV nmx_c2* nmx_w500_fft(const nmx_c2* in, nmx_c2* a, nmx_c2* b, const NmxW500Tw& T, int lane) {
  nmx_c2 v[10];
  if (lane < 50) {
    // stage 1: R = 10, Ns = 1: in[j + 50 r] -> a[10 j + r]
#pragma unroll
    for (int r = 0; r < 10; ++r) v[r] = in[lane + 50 * r];
    nmx_dft10_c2<DIR>(v);
    nmx_c2* o = a + 10 * lane;
#pragma unroll
    for (int r = 0; r < 10; ++r) o[r] = v[r];
  }
  NMX_WAVE_FENCE();
  if (lane < 50) {
    // stage 2: R = 10, Ns = 10: a[j + 50 r] * w^(5 k r) -> b[100 q + k + 10 r],  q = j / 10, k = j % 10
#pragma unroll
    for (int r = 0; r < 10; ++r) v[r] = a[lane + 50 * r];
#pragma unroll
    for (int r = 1; r < 10; ++r) v[r] = nmx_cmul(v[r], nmx_twd<DIR>(T.s2[r - 1]));
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
      const nmx_c2* src = b + lane + 50 * h;
      nmx_c2 x0 = src[0], x1 = src[100], x2 = src[200], x3 = src[300], x4 = src[400];
      const nmx_c2* w = h ? T.s3b : T.s3a;
      x1 = nmx_cmul(x1, nmx_twd<DIR>(w[0]));
      x2 = nmx_cmul(x2, nmx_twd<DIR>(w[1]));
      x3 = nmx_cmul(x3, nmx_twd<DIR>(w[2]));
      x4 = nmx_cmul(x4, nmx_twd<DIR>(w[3]));
      nmx_dft5_c2<DIR>(x0, x1, x2, x3, x4);
      nmx_c2* o = a + lane + 50 * h;
      o[0] = x0; o[100] = x1; o[200] = x2; o[300] = x3; o[400] = x4;
    }
  }
  NMX_WAVE_FENCE();
  return a;
}

// Hilbert envelope of one length-1000 series per WAVE (same math as nmx_hilbert_item, even-W branch).
// LDS per wave: a[500] + b[501] complex + ys[1000] floats.
#define NMX_W500_LDS_FLOATS (1000 + 1004 + 1000)
NMX_DEV void nmx_hilbert_w500_item(const NmxHilbertArgs& A, long long item, float* smem) {
  const int lane = NMX_TID;
  nmx_c2* a = (nmx_c2*)smem;
  nmx_c2* b = (nmx_c2*)(smem + 1000);
  float* ys = smem + 2004;
  const float* src = A.y + item * 1000;
  float* dst = A.env + item * 1000;
  NmxW500Tw T;
  nmx_w500_load_tw(T, A.hil_r.tw, lane);
  {
    float* pk = (float*)b;   // packed complex: (x[2i], x[2i+1])
    nmx_stage_row(src, 1000, [=](int i, float v) { pk[i] = v; ys[i] = v; });
  }
  NMX_WAVE_FENCE();
  const nmx_c2* Z = nmx_w500_fft<-1>(b, a, b, T, lane);   // = a
  // Hermitian half Y[0..500] of the length-1000 real transform -> b (Z = a stays intact)
  const float2* Zf = (const float2*)Z;
  float2* Yb = (float2*)b;
  for (int k = lane; k <= 500; k += 64) Yb[k] = nmx_rfft_bin(Zf, A.hil_r.twr, 500, k);
  NMX_WAVE_FENCE();
  // Z'[k] of the half-length inverse of -i Y (DC and Nyquist dropped) -> a
  float2* Pre = (float2*)a;
  for (int k = lane; k < 500; k += 64) {
    const float2 yk = Yb[k], yn = Yb[500 - k];
    const float2 xk = k == 0 ? make_float2(0.f, 0.f) : make_float2(yk.y, -yk.x);
    const float2 xn = k == 0 ? make_float2(0.f, 0.f) : make_float2(yn.y, -yn.x);
    Pre[k] = nmx_irfft_pre(xk, xn, A.hil_r.twr[k]);
  }
  NMX_WAVE_FENCE();
  // in = a, first output buffer must differ from the input: a -> b -> a -> b
  const float* ht = (const float*)nmx_w500_fft<+1>(a, b, a, T, lane);
  const float invW = 1.f / 1000.f;
  for (int i = lane; i < 1000; i += 64) {
    const float re = ys[i], im = ht[i] * invW;
    dst[i] = sqrtf(re * re + im * im);
  }
}
#endif

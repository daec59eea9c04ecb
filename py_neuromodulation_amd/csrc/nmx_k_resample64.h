// nmx_k_resample64.h -- the stand-alone Resampler in FLOAT64 (nmx_resample_f64), any window length.
//
// Reference: processing/resample.py:42-60 -> mne.filter.resample(x.astype(float64), up = new / old, down = 1): FFT method,
// boxcar window, npad "auto", reflect_limited padding; restated in oracle/mne_restated.py::resample (MNE itself is absent:
// PARITY UNPINNED against it).  The reference's own tests resample 10 s at 4 kHz in one call (tests/test_nm_resample.py:
// 8-47); inside a plan the resampler is nmx_k_resample.h on fp32 windows that fit LDS.  Here every step is a pass over
// HBM in float64, one thread per complex point:
//   PAD     z[i] = reflect_limited extension of the window, i < n_pad = 2^ceil(log2(W + 2 min(W / 8, 100)))
//   PASS    one radix-2 Stockham (autosort, decimation in frequency) stage of a power-of-two transform; log2(n) launches
//   MAP     the n_new-point Hermitian spectrum from the n_pad-point one: bins kept / zero-extended, the Nyquist bin of the
//           shorter length doubled (down-sampling) or halved (up-sampling), the imaginary parts of the DC and the n_new / 2
//           bins dropped as a C2R transform does
//   n_new a power of two: PASSes with the conjugate roots; any other n_new: Bluestein -- the length-n_new inverse as a
//           circular convolution of length L = 2^ceil(log2(2 n_new - 1)) with the chirp exp(-i pi m^2 / n_new) (CHIRP, MUL),
//           three power-of-two transforms
//   OUT     y[i] = ratio / n_new * Re(...)[crop_l + i], i < W_new = round(ratio W)
#pragma once

#include "nmx_device.h"

struct NmxCplx64 { double x, y; };

enum { NMX_RS64_PAD = 0, NMX_RS64_MAP = 1, NMX_RS64_CHIRP = 2, NMX_RS64_MUL = 3, NMX_RS64_OUT = 4 };

struct NmxResample64Args {
  const double* x;       // [C][ldx] raw windows
  long long ldx;
  double* y;             // [C][ldy] resampled
  long long ldy;
  NmxCplx64* a;          // [C][ld]  work
  NmxCplx64* b;          // [C][ld]  work
  NmxCplx64* chirp;      // [L]      spectrum of the Bluestein chirp (one row), or NULL
  long long ld;          // row stride of a / b in complex points
  int C;
  long long W, W_new, n_pad, pad_l, n_new, crop_l;
  long long L;           // Bluestein length (0: n_new is a power of two)
  long long nyq_bin;     // -1: none
  double nyq_scale, scale;
};

NMX_DEV void nmx_sincospi64(double a, double* s, double* c) {
#ifdef NMX_HOST_EMU
  *s = sin(3.141592653589793238462643383279502884 * a);
  *c = cos(3.141592653589793238462643383279502884 * a);
#else
  sincospi(a, s, c);
#endif
}
NMX_DEV NmxCplx64 nmx_cmul64(NmxCplx64 p, NmxCplx64 q) { return NmxCplx64{p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x}; }
// exp(+i pi m^2 / M): m^2 reduced mod 2 M in integers first (m < 2^31)
NMX_DEV NmxCplx64 nmx_chirp64(long long m, long long M) {
  const long long r = (long long)(((unsigned long long)m * (unsigned long long)m) % (unsigned long long)(2 * M));
  double s, c;
  nmx_sincospi64((double)r / (double)M, &s, &c);
  return NmxCplx64{c, s};
}

// One butterfly of a Stockham stage: `src` -> `dst`, rows of n points, this stage of length ns (n, n / 2, ... 2), stride
// st = n / ns; t in [0, n / 2).  sign = -1 forward, +1 inverse (no scaling).
NMX_DEV void nmx_rs64_pass(const NmxCplx64* src, NmxCplx64* dst, long long ld, long long n, long long ns, long long st, int sign,
                           int c, long long t) {
  if (t >= (n >> 1)) return;
  const long long p = t / st, q = t - p * st, m = ns >> 1;
  const NmxCplx64* s = src + (long long)c * ld;
  NmxCplx64* d = dst + (long long)c * ld;
  const NmxCplx64 u = s[q + st * p], v = s[q + st * (p + m)];
  double sn, cs;
  nmx_sincospi64(2.0 * (double)p / (double)ns, &sn, &cs);
  const NmxCplx64 w{cs, sign < 0 ? -sn : sn};
  d[q + st * (2 * p)] = NmxCplx64{u.x + v.x, u.y + v.y};
  d[q + st * (2 * p + 1)] = nmx_cmul64(NmxCplx64{u.x - v.x, u.y - v.y}, w);
}

// The element-wise stages; `src` / `dst` are a / b in whichever order the transforms left them.
NMX_DEV void nmx_rs64_elem(const NmxResample64Args& A, int mode, const NmxCplx64* src, NmxCplx64* dst, int c, long long i) {
  if (mode == NMX_RS64_PAD) {
    if (i >= A.n_pad) return;
    const double* x = A.x + (long long)c * A.ldx;
    const long long W = A.W, j = i - A.pad_l;
    double v;
    if (j < 0) v = (-j <= W - 1) ? 2.0 * x[0] - x[-j] : 0.0;
    else if (j < W) v = x[j];
    else {
      const long long r = j - (W - 1);
      v = (r <= W - 1) ? 2.0 * x[W - 1] - x[W - 1 - r] : 0.0;
    }
    dst[(long long)c * A.ld + i] = NmxCplx64{v, 0.0};
  } else if (mode == NMX_RS64_MAP) {
    // bin k of the n_new-point spectrum (k > n_new / 2: the conjugate of bin n_new - k); Bluestein: times the chirp, zeros to L
    const long long M = A.n_new, n_out = A.L ? A.L : M;
    if (i >= n_out) return;
    NmxCplx64 v{0.0, 0.0};
    if (i < M) {
      const long long k = i <= (M >> 1) ? i : M - i;
      if (k <= (A.n_pad >> 1)) {
        v = src[(long long)c * A.ld + k];
        if (k == A.nyq_bin) { v.x *= A.nyq_scale; v.y *= A.nyq_scale; }
      }
      if (k == 0 || ((M & 1) == 0 && k == (M >> 1))) v.y = 0.0;
      if (i > (M >> 1)) v.y = -v.y;
      if (A.L) v = nmx_cmul64(v, nmx_chirp64(i, M));
    }
    dst[(long long)c * A.ld + i] = v;
  } else if (mode == NMX_RS64_CHIRP) {
    // b[m] = exp(-i pi m^2 / M) for |m| < M laid out circularly on L points (one row)
    if (i >= A.L) return;
    const long long M = A.n_new, m = i < M ? i : (A.L - i < M ? A.L - i : -1);
    NmxCplx64 v{0.0, 0.0};
    if (m >= 0) { v = nmx_chirp64(m, M); v.y = -v.y; }
    dst[i] = v;
  } else if (mode == NMX_RS64_MUL) {
    if (i >= A.L) return;
    NmxCplx64* p = dst + (long long)c * A.ld + i;
    *p = nmx_cmul64(*p, A.chirp[i]);
  } else {
    if (i >= A.W_new) return;
    const long long n = A.crop_l + i;
    NmxCplx64 v = src[(long long)c * A.ld + n];
    if (A.L) v = nmx_cmul64(v, nmx_chirp64(n, A.n_new));
    A.y[(long long)c * A.ldy + i] = v.x * A.scale;
  }
}

// nmx_k_timeosc_w510.h -- kernel A for 510-sample transforms (17 ms at 30 kHz: BASELINE config 5), ONE WAVE per
// (window, channel): FFT over the last 510 samples of the window and STFT with 510-sample segments (hop 255, even
// boundary), band MEANS only.  Reference arithmetic: features/oscillatory.py:90-119, 215-250 (see nmx_k_timeosc.h).
//
// 510 = 2 . 3 . 5 . 17 has pairwise coprime factors: the Good-Thomas prime-factor transform needs NO twiddles and runs
// IN PLACE -- every small DFT reads and writes the same positions of the 510-point buffer in LDS (tools/model_pfa510.py):
//   position p has coordinates c_d = p (N / N_d)^-1 mod N_d;  the DFT along dimension d runs over (g + c N / N_d) mod N;
//   afterwards position p holds X[k], k = the Chinese-remainder index of p's coordinates (table pos_of_k on the host).
// Two REAL sequences ride in one complex transform (z = a + i b; A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = -i (Z[k] -
// conj Z[N-k]) / 2): the FFT window and the STFT segments are paired up.  Phases per transform:
//   (2 x 5)  51 lanes, 10 points in registers each: two 5-point DFTs and five butterflies
//   3        170 groups over three rounds of 64 lanes
//   17       30 groups; TWO transforms side by side (lanes 0..29 and 32..61).  A 17-point DFT in the symmetric form
//            y_h, y_{17-h} = x_0 + SA -+ i SB,  SA = sum_n (x_n + x_{17-n}) cos(2 pi h n / 17),  SB = sum_n (x_n - x_{17-n}) sin(..)
//            with compile-time coefficients: 128 packed fused multiply-adds for the sixteen outputs.
// The generic workgroup kernel spends 5.5 ms per 1024 hops x 512 channels on these transforms (255-point complex through
// run-time stage tables, 128 threads and ~8 barriers per transform).
// LDS per wave: window (<= 1024 floats) + two 510-point complex buffers = 12 KB.
#pragma once

#include "nmx_k_timeosc_w1000.h"

#ifndef NMX_HOST_EMU

// the window (its length rounded up to 16 floats + 16: the extension reads stay inside) + two 512-point complex buffers:
// 10.1 KB per wave for 512-sample windows, 12 KB for 1024
#define NMX_TO510_XS_FLOATS(W) ((((W) + 15) & ~15) + 16)
#define NMX_TO510_LDS_FLOATS(W) (NMX_TO510_XS_FLOATS(W) + 2 * 1024)
// table layout (unsigned short): g10[51] (padded to 64) | g3[170] (padded to 192) | g17[30] (padded to 32) | pos_of_k[510] (padded to 512)
#define NMX_TO510_TAB_G10 0
#define NMX_TO510_TAB_G3 64
#define NMX_TO510_TAB_G17 256
#define NMX_TO510_TAB_POSK 288
#define NMX_TO510_TAB_N 800

static inline bool nmx_timeosc_w510_ok(const NmxTimeOscArgs& A, const unsigned short* tab) {
  if (!tab || A.W < 510 || A.W > 1024 || A.n_bands > 8 || A.welch.enabled) return false;
  auto mean_only = [](const NmxOsc& O) { return !O.complex_full && O.estimators == NMXD_EST_MEAN && !O.return_spectrum; };
  if (A.fft.enabled && !(mean_only(A.fft) && A.fft.n == 510 && A.fft.k_lo >= 1 && A.fft.k_hi <= 255)) return false;
  auto short_ok = [](const NmxOsc& O) {   // segments of at most 64 samples: direct DFT per (segment, bin)
    return O.estimators == NMXD_EST_MEAN && !O.return_spectrum && O.n <= 64 && O.n >= 2;
  };
  if (A.stft.enabled && !short_ok(A.stft) &&
      !(mean_only(A.stft) && A.stft.n == 510 && A.stft.step == 255 && A.stft.half == 255 &&
        A.stft.nseg >= 1 && A.stft.nseg <= 8 && A.stft.k_lo >= 1 && A.stft.k_hi <= 255)) return false;
  return (A.fft.enabled && A.fft.n == 510) || (A.stft.enabled && A.stft.n == 510);
}

NMX_DEV int nmx_mod510(int p) { return p >= 510 ? p - 510 : p; }

// phase (2 x 5): lane < 51 owns the positions (g + 255 c2 + 102 c5) mod 510
NMX_DEV void nmx_pfa510_phase10(float2* buf, const unsigned short* tab, int lane) {
  if (lane >= 51) return;
  const int g = tab[NMX_TO510_TAB_G10 + lane];
  int pos[2][5];
  float2 v[2][5];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      pos[a][b] = nmx_mod510(nmx_mod510(g + 255 * a) + 102 * b);
      v[a][b] = buf[pos[a][b]];
    }
#pragma unroll
  for (int a = 0; a < 2; ++a) nmx_dft5<-1>(v[a][0], v[a][1], v[a][2], v[a][3], v[a][4]);
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const float2 s = nmx_cadd(v[0][b], v[1][b]), d = nmx_csub(v[0][b], v[1][b]);
    buf[pos[0][b]] = s;
    buf[pos[1][b]] = d;
  }
}
// phase 3: 170 groups (g + 170 c) mod 510 over three rounds
NMX_DEV void nmx_pfa510_phase3(float2* buf, const unsigned short* tab, int lane) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int idx = lane + 64 * r;
    if (idx >= 170) continue;
    const int g = tab[NMX_TO510_TAB_G3 + idx];
    const int p0 = g, p1 = nmx_mod510(g + 170), p2 = nmx_mod510(g + 340);
    const float2 a0 = buf[p0], a1 = buf[p1], a2 = buf[p2];
    const float s = -0.86602540378443865f;   // forward
    const float2 t = nmx_cadd(a1, a2), d = nmx_csub(a1, a2);
    const float2 u = make_float2(a0.x - 0.5f * t.x, a0.y - 0.5f * t.y);
    const float2 v = make_float2(-s * d.y, s * d.x);
    buf[p0] = nmx_cadd(a0, t);
    buf[p1] = nmx_cadd(u, v);
    buf[p2] = nmx_csub(u, v);
  }
}
// 17-point DFT (forward) of x[0..16] in place, symmetric form
NMX_DEV void nmx_dft17_fwd(float2* x) {
  constexpr float C[9] = {1.f, 0.93247222940435581f, 0.73900891722065920f, 0.44573835577653826f, 0.09226835946330200f,
                          -0.27366299007208286f, -0.60263463637925638f, -0.85021713572961420f, -0.98297309968390178f};
  constexpr float S[9] = {0.f, 0.36124166618715292f, 0.67369564364655721f, 0.89516329135506234f, 0.99573417629503447f,
                          0.96182564317281904f, 0.79801722728023949f, 0.52643216287735580f, 0.18374951781657034f};
  float2 a[9], b[9];
  float2 s0 = x[0];
#pragma unroll
  for (int n = 1; n <= 8; ++n) {
    a[n] = nmx_cadd(x[n], x[17 - n]);
    b[n] = nmx_csub(x[n], x[17 - n]);
    s0 = nmx_cadd(s0, a[n]);
  }
  const float2 x0 = x[0];
  x[0] = s0;
#pragma unroll
  for (int h = 1; h <= 8; ++h) {
    float sax = x0.x, say = x0.y, sbx = 0.f, sby = 0.f;
#pragma unroll
    for (int n = 1; n <= 8; ++n) {
      const int e = (h * n) % 17;                          // compile-time after unrolling
      const float cs = e <= 8 ? C[e] : C[17 - e];
      const float sn = e <= 8 ? S[e] : -S[17 - e];
      sax += a[n].x * cs; say += a[n].y * cs;
      sbx += b[n].x * sn; sby += b[n].y * sn;
    }
    // forward: y_h = x0 + SA - i SB,  y_{17-h} = x0 + SA + i SB
    x[h] = make_float2(sax + sby, say - sbx);
    x[17 - h] = make_float2(sax - sby, say + sbx);
  }
}
// phase 17: 30 groups (g + 30 c) mod 510; lanes 0..29 work on bufA, lanes 32..61 on bufB (when `two`)
NMX_DEV void nmx_pfa510_phase17(float2* bufA, float2* bufB, bool two, const unsigned short* tab, int lane) {
  const int li = lane & 31, tr = lane >> 5;
  if (li >= 30 || (tr && !two)) return;
  float2* buf = tr ? bufB : bufA;
  const int g = tab[NMX_TO510_TAB_G17 + li];
  int pos[17];
  float2 x[17];
#pragma unroll
  for (int c = 0; c < 17; ++c) {
    pos[c] = nmx_mod510(g + 30 * c);
    x[c] = buf[pos[c]];
  }
  nmx_dft17_fwd(x);
#pragma unroll
  for (int c = 0; c < 17; ++c) buf[pos[c]] = x[c];
}

// short STFT segments: one segment per lane and round, NBK bins (k_lo + b0 ...) from one pass over its windowed samples
template <int NB, int NBK>
NMX_DEV void nmx_w510_short_stft(const NmxOsc& OS, const float* xs, const float2* rootsL, const float* winL, int W, int nb,
                                 int b0, int lane, NmxBandAcc<NB>& acc_s) {
  const int N = OS.n, h = OS.half;
  const float lscale = OS.log_transform ? OS.log10_scale : 0.f;
  for (int sgi = lane; sgi < OS.nseg; sgi += 64) {
    const int s0 = sgi * OS.step;
    float re[NBK], im[NBK];
    int m[NBK];
#pragma unroll
    for (int b = 0; b < NBK; ++b) { re[b] = 0.f; im[b] = 0.f; m[b] = 0; }
    for (int i = 0; i < N; ++i) {
      const int e = s0 + i;
      float v;
      if (e < h) v = xs[h - e];
      else if (e < h + W) v = xs[e - h];
      else if (e < 2 * h + W) v = xs[W - 2 - (e - h - W)];
      else v = 0.f;
      v *= winL[i];
#pragma unroll
      for (int b = 0; b < NBK; ++b) {
        const float2 r = rootsL[m[b]];
        re[b] += v * r.x;
        im[b] += v * r.y;
        m[b] += OS.k_lo + b0 + b;
        if (m[b] >= N) m[b] -= N;
      }
    }
#pragma unroll
    for (int b = 0; b < NBK; ++b) {
      const float pw = re[b] * re[b] + im[b] * im[b];
      const float v = OS.log_transform ? nmx_log10_half_fast(pw) + lscale : sqrtf(pw) * OS.scale;
      acc_s.add(OS, nb, OS.k_lo + b0 + b, v);
    }
  }
}

#ifdef NMX_W510_PROFILE
#define NMX_W5P(i) { const long long t_ = clock64(); w5p[i] += t_ - w5l; w5l = t_; }
#else
#define NMX_W5P(i)
#endif
template <int NB>
NMX_DEV void nmx_timeosc_w510_item(const NmxTimeOscArgs& A0, const unsigned short* tab, int w, int c, float* smem) {
#ifdef NMX_W510_PROFILE
  long long w5p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w5l = clock64();
#endif
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  const int lane = (int)(threadIdx.x & 63);
  // The plan is re-read (s_load) where a phase needs it instead of living in scalar registers from the top of the kernel:
  // the pointer is laundered between the phases (kept alive, the plan spilled 260 scalars to VGPR lanes and every use paid
  // a v_readlane / v_writelane -- vector instructions of an issue-bound kernel)
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* nmx_karg_p;   // (A0 lives in the kernel-argument segment)
  nmx_karg_p Aq = (nmx_karg_p)(unsigned long long)&A0;
#define NMX_W510_RELOAD() asm volatile("" : "+s"(Aq))
#define A (*(const NmxTimeOscArgs*)Aq)
  NMX_W510_RELOAD();
  const int W = A.W;
  float* xs = smem;                              // [W] the window, natural order
  float2* bufA = (float2*)(smem + NMX_TO510_XS_FLOATS(W));          // [512] (510 used)
  float2* bufB = (float2*)(smem + NMX_TO510_XS_FLOATS(W) + 1024);   // [512]
  float* out_row = A.out + (long long)w * A.n_outputs;
  const int nb = A.n_bands;

  // time domain on packed arithmetic (nmx_k_td.h: 200 vector instructions where the scalar formulation of nmx_k_scan.h
  // takes 600); its window sum is also the NaN / infinity test of the window
  bool fast = false;
  if (nmx_td_ok(A)) {
    NmxTdRegs Rt;
    nmx_td_load(A, w, c, Rt);
    fast = nmx_td_emit(A, w, c, Rt, nullptr);
    if (fast) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * (lane + 64 * k) < W) ((nmx_f4*)xs)[lane + 64 * k] = Rt.x[k];
    }
  }
  if (!fast) {   // odd window lengths, a NaN or an infinity in the window: cleaning loads, scalar formulation
    NmxScanRegs R;
    nmx_scan_load(A, w, c, R);
    R.sum = 0.f;
    if (A.features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) nmx_scan_emit(A, w, c, R);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n0 = 4 * (lane + 64 * k);
      if (n0 < W) ((nmx_f4*)xs)[lane + 64 * k] = nmx_f4{R.x[k][0], R.x[k][1], R.x[k][2], R.x[k][3]};   // (W % 4 == 0 or the tail reads zeros)
    }
  }
  NMX_WAVE_FENCE();
  NMX_W5P(0)   // load + time-domain features

  // the real sequences, in order: [FFT window] + STFT segments 0 .. nseg - 1
  const bool stft_long = A.stft.enabled && A.stft.n == 510;
  const int n_fft = A.fft.enabled ? 1 : 0, n_seq = n_fft + (stft_long ? A.stft.nseg : 0);
#define OS (A.stft)
  const int h = OS.half;
  // element i of STFT segment sg (branch-free): position e of the evenly extended, zero padded window is x[|e - h|] up to
  // the window's end, x[2 W - 2 + h - e] in the right extension, 0 beyond; wv = the segment window's coefficient
  auto seg_sample = [&](int sg, int i, float wv) -> float {
    const int e = sg * 255 + i;
    const int d = e - h;
    const int idx = e < h + W ? (d < 0 ? -d : d) : 2 * W - 2 - d;
    const bool valid = e < 2 * h + W;
    return valid ? xs[valid ? idx : 0] * wv : 0.f;
  };
  NmxBandAcc<NB> acc_f, acc_s;
  acc_f.clear();
  acc_s.clear();
  auto bins = [&](const float2* Z, int qa, int qb) {   // band contributions of the sequences qa (real part), qb (imaginary; -1: none)
    const int k_lo = (A.fft.enabled && stft_long) ? (A.fft.k_lo < OS.k_lo ? A.fft.k_lo : OS.k_lo) : (A.fft.enabled ? A.fft.k_lo : OS.k_lo);
    const int k_hi = (A.fft.enabled && stft_long) ? (A.fft.k_hi > OS.k_hi ? A.fft.k_hi : OS.k_hi) : (A.fft.enabled ? A.fft.k_hi : OS.k_hi);
    for (int k = k_lo + lane; k < k_hi; k += 64) {
      const float2 zk = Z[tab[NMX_TO510_TAB_POSK + k]], zn = Z[tab[NMX_TO510_TAB_POSK + 510 - k]];
      const float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y);
      const float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x);
      for (int s = 0; s < 2; ++s) {
        const int q = s ? qb : qa;
        if (q < 0) continue;
        const float re = s ? bx : ax, im = s ? by : ay;
        const float pw = re * re + im * im;
        if (q < n_fft) {
          if (k >= A.fft.k_lo && k < A.fft.k_hi)
            acc_f.add(A.fft, nb, k, A.fft.log_transform ? nmx_log10_half_fast(pw) : sqrtf(pw));
        } else if (k >= OS.k_lo && k < OS.k_hi) {
          acc_s.add(OS, nb, k, OS.log_transform ? nmx_log10_half_fast(pw) + OS.log10_scale : sqrtf(pw) * OS.scale);
        }
      }
    }
  };
  for (int q0 = 0; q0 < n_seq; q0 += 4) {
    NMX_W510_RELOAD();
    // up to two complex transforms (four real sequences) per round: bufA = (q0, q0 + 1), bufB = (q0 + 2, q0 + 3)
    const bool two = q0 + 2 < n_seq;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // (entries 510 and 511 of the 512-point buffers take whatever lanes 62 / 63 compute: never read)
      const int i = lane + 64 * r;
      const float wv = stft_long ? OS.win[i < 510 ? i : 509] : 0.f;
      auto seq = [&](int q) -> float {   // sequence q: the FFT window first, then the segments (q wave-uniform)
        if (q >= n_seq) return 0.f;
        return q < n_fft ? xs[W - 510 + i] : seg_sample(q - n_fft, i, wv);
      };
      bufA[i] = make_float2(seq(q0), seq(q0 + 1));
      if (two) bufB[i] = make_float2(seq(q0 + 2), seq(q0 + 3));
    }
    NMX_WAVE_FENCE();
    NMX_W5P(1)   // fill
    nmx_pfa510_phase10(bufA, tab, lane);
    if (two) nmx_pfa510_phase10(bufB, tab, lane);
    NMX_WAVE_FENCE();
    NMX_W5P(2)
    nmx_pfa510_phase3(bufA, tab, lane);
    if (two) nmx_pfa510_phase3(bufB, tab, lane);
    NMX_WAVE_FENCE();
    NMX_W5P(3)
    nmx_pfa510_phase17(bufA, bufB, two, tab, lane);
    NMX_WAVE_FENCE();
    NMX_W5P(4)
    NMX_W510_RELOAD();
    bins(bufA, q0, q0 + 1 < n_seq ? q0 + 1 : -1);
    if (two) bins(bufB, q0 + 2, q0 + 3 < n_seq ? q0 + 3 : -1);
    NMX_WAVE_FENCE();
    NMX_W5P(5)
  }
  if (A.stft.enabled && !stft_long) {
    // short segments (17 samples at 30 kHz: dozens of segments per window): ONE SEGMENT per lane and round, all its
    // bins (up to eight at a time) from one pass over the windowed samples; the roots exp(-2 pi i m / N) and the
    // window sit in LDS (the generic kernel evaluated a sincospi per term and (segment, bin) pair)
    const int N = OS.n, nbins = OS.k_hi - OS.k_lo;
    float2* rootsL = bufB;                 // [N]
    float* winL = (float*)(bufB + 64);     // [N]
    NMX_WAVE_FENCE();
    if (lane < N) {
      float sn, cs;
      sincospif(-2.f * (float)lane / (float)N, &sn, &cs);
      rootsL[lane] = OS.complex_full ? OS.fft.tw[lane] : make_float2(cs, sn);   // (odd N: the plan's exactly rounded table)
      winL[lane] = OS.win[lane];
    }
    NMX_WAVE_FENCE();
    for (int b0 = 0; b0 < nbins; b0 += 8) {
      const int nbk = nbins - b0 < 8 ? nbins - b0 : 8;
      switch (nbk) {   // compile-time bin count: no per-term predicates
        case 1: nmx_w510_short_stft<NB, 1>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        case 2: nmx_w510_short_stft<NB, 2>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        case 3: nmx_w510_short_stft<NB, 3>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        case 4: nmx_w510_short_stft<NB, 4>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        case 5: nmx_w510_short_stft<NB, 5>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        case 6: nmx_w510_short_stft<NB, 6>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        case 7: nmx_w510_short_stft<NB, 7>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
        default: nmx_w510_short_stft<NB, 8>(OS, xs, rootsL, winL, W, nb, b0, lane, acc_s); break;
      }
    }
  }
  NMX_W510_RELOAD();
  if (A.fft.enabled) acc_f.emit(A.fft, nb, 1, out_row, c, lane);
  if (A.stft.enabled) acc_s.emit(OS, nb, OS.nseg, out_row, c, lane);
  NMX_W5P(6)
#ifdef NMX_W510_PROFILE
  if (lane == 0 && c == 7 && (w == 3 || w == 600))
    printf("[w510 w=%d] cycles: load %lld fill %lld p10 %lld p3 %lld p17 %lld bins %lld emit %lld\n", w, w5p[0], w5p[1], w5p[2],
           w5p[3], w5p[4], w5p[5], w5p[6]);
#endif
}
#undef A
#undef OS
#undef NMX_W510_RELOAD

#endif

// nmx_k_bank_w64.h -- kernel B', the fast FIR-bank path: ONE WAVE per (window, channel),
// circular-convolution length M = 2048 (half-length complex transform n = 1024 = 64 lanes x 16).
//
// CDNA4 mapping
//   * every lane owns 16 complex points in VGPRs; a 1024-point transform is three register
//     passes (radix 16, 16, 4) separated by TWO LDS exchanges instead of five LDS round trips,
//     all twiddles live in VGPRs (loaded once per workgroup, reused for 1 forward + n_filters
//     inverse transforms), no barriers across waves (single-wave workgroup).
//   * LDS reads are always lane-consecutive ds_read_b64 (Stockham addressing); the only strided
//     write (first pass, 16 consecutive points per lane) goes to a buffer padded by one point
//     per 16 -> conflict-free ds_write_b64.  The exchange buffer is reused in place (a wave
//     reads all 16 points before it writes any).
//   * forward transform: window HBM -> registers (float2, lane-consecutive) -> pass A directly;
//     its result Z (8 KiB) stays in LDS for all filters.
//   * per filter the forward "split", the multiplication by the REAL tap spectrum H and the
//     inverse "unsplit" of the half-length real-FFT trick collapse algebraically to
//         Z'[k] = A_k Z[k] + i B_k conj(Z[n-k]),   A_k = (a+b) - (a-b) sin(2 pi k / M),
//                                                   B_k = (a-b) cos(2 pi k / M),  a = H[k], b = H[n-k]
//     i.e. 4 flops per point with two real tables; the Hermitian spectrum X is never formed.
//   * epilogues from registers: band-pass activity = two-pass variance over the tail samples
//     with wave reductions (nothing written to LDS); filtered series for sharp waves / bursts /
//     notch are stored lane-consecutively to HBM.
// The forward transform Z stays in VGPRs (the conjugate partner Z[n-k] is one cross-lane read from
// lane 64 - l), so the only LDS tile is the 8.5 KiB exchange buffer -> up to 16 waves per CU.
#pragma once

#include "nmx_k_bank.h"
#include "nmx_k_fft500.h"
#include "nmx_k_sharpwave.h"

#ifdef NMX_HOST_EMU
#define NMX_UNROLL
#else
#define NMX_UNROLL _Pragma("unroll")
#endif

#ifdef NMX_HOST_EMU
#define NMX_SCHED_FENCE() ((void)0)
#else
// keep the scheduler from hoisting all 16 points' loads above the first use (register pressure)
#define NMX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// -DNMX_BANK_PROFILE: per-phase cycle counters (s_memtime) of one wave, printed for two items -- where a wave's
// time goes inside the filter loop (tools/exp_variants.sh with NMX_EXTRA_CXXFLAGS; never in the product build)
#if defined(NMX_BANK_PROFILE) && !defined(NMX_HOST_EMU)
#define NMX_PROF_DECL long long pf_t = clock64(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NMX_PROF(i) { const long long t_ = clock64(); pf_acc[i] += t_ - pf_t; pf_t = t_; }
#define NMX_PROF_PRINT(w, c)                                                                                          \
  if ((w) == 5 && ((c) == 3 || (c) == 40) && (threadIdx.x & 63) == 0)                                                  \
    printf("bank profile w=%d c=%d: load+fwd %lld | spectral+passA %lld | passB load+dft %lld | passB store %lld | "  \
           "passC %lld | variance %lld | series stores %lld | partners %lld (cycles, whole item, %d filters)\n", (w), (c), pf_acc[0],   \
           pf_acc[1], pf_acc[2], pf_acc[3], pf_acc[4], pf_acc[5], pf_acc[7], pf_acc[6], A.n_filters);
#else
#define NMX_PROF_DECL
#define NMX_PROF(i)
#define NMX_PROF_PRINT(w, c)
#endif

// Cache policy of the filtered-series stores: non-temporal (aux bit 1).  The series stream is several times
// the input and is read again only by later kernels; written with the default policy it evicts the
// overlapping input windows from L2 before the next hop of the same wave re-reads them.  Measured on the
// channel-pair kernel reading the ring directly (no notch stage): HBM fetch 0.40 -> 0.07 GB per launch,
// duration unchanged (the kernel is VALU-bound).  Behind a notch stage every window is a distinct buffer
// and the 1.05 GB fetch is compulsory either way.
#ifndef NMX_SERIES_STORE_AUX
#define NMX_SERIES_STORE_AUX 2
#endif
#define NMX_W64_N 1024
#define NMX_W64_E 16
// register index of point l + 64 j after pass C (v[4 t + r] = y[l + 64 t + 256 r], j = t + 4 r)
#define NMX_J2I(j) (4 * ((j) & 3) + ((j) >> 2))

struct NmxBankW64Args {
  NmxBankArgs b;          // shared description (x, strides, filters, cols, outputs ...)
  // per filter, k < n = M/2, th_k = 2 pi k / M, a = H[k], b = H[n-k] (real spectrum of the taps):
  const float* Hs[NMX_MAX_FILTERS_DEV];   // A_k = (a + b) - (a - b) sin(th_k)
  const float* Hd[NMX_MAX_FILTERS_DEV];   // B_k = (a - b) cos(th_k)
  float* yb_out;          // burst bands: filtered series [n_windows][C][Bb][W] (Hilbert kernel input)
  const float* twl;       // NMX_W64_TWL_FLOATS floats: per-lane twiddles of passes B and C (persistent kernel)
  const float* hc;        // M = 1536 channel-pair path (nmx_k_bank_w64c.h): [n_filters][12][64] pairs of the REAL spectrum in
                          // register order; twc = [24][64] complex pass-A twiddles, then [8][8] complex exp(-2 pi i a b / 64)
  const float* twc;
  int pair_m;             // 1536 (nmx_k_bank_w64c.h) or 1024 (nmx_k_bank_w64d.h: hc = [n_filters][8][64] pairs in natural order,
                          // twiddles = twl)
  const float* tw2;       // M = 4096 path (nmx_k_bank_w64x2.h): [1024] complex exp(-2 pi i k / 2048); Hs[f] then holds the
                          // INTERLEAVED (A_k, B_k) table of filter f, 2048 pairs
  int off_Z, off_X, off_red, lds_floats;
};

#ifdef NMX_HOST_EMU
#define NMX_LANES 64
#define NMX_LANE_LOOP for (int l = 0; l < 64; ++l)
#define NMX_LI l
#define NMX_WSYNC() ((void)0)
#else
#define NMX_LANES 1
#define NMX_LANE_LOOP for (int l = (int)(threadIdx.x & 63), l_once_ = 0; l_once_ < 1; ++l_once_)
#define NMX_LI 0
#define NMX_WSYNC() NMX_WAVE_FENCE()
#endif

// ---- buffer addressing (device) ---------------------------------------------------------------
// Rows are read / written through raw buffer descriptors whose num_records is the row length:
// out-of-range lanes read 0 and their stores are dropped BY THE HARDWARE, so the zero padding of the
// window and the ragged end of a row cost no compares, no exec-mask branches and no 64-bit address
// arithmetic (the offsets 512 r are instruction immediates).
#ifndef NMX_HOST_EMU
typedef __amdgpu_buffer_rsrc_t nmx_rsrc;
NMX_DEV nmx_rsrc nmx_make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
#endif

// 4-point DFT, DIR = -1 forward / +1 inverse
template <int DIR>
NMX_DEV void nmx_dft4(nmx_c2& a0, nmx_c2& a1, nmx_c2& a2, nmx_c2& a3) {
  const nmx_c2 t0 = nmx_cadd(a0, a2), t1 = nmx_csub(a0, a2), t2 = nmx_cadd(a1, a3);
#if defined(NMX_LDS_ASM) && !defined(NMX_HOST_EMU)
  const nmx_c2 d = nmx_csub(a1, a3);   // the +-i of the odd outputs rides on the adds (nmx_device.h)
  a0 = nmx_cadd(t0, t2);
  a1 = nmx_add_ib<DIR>(t1, d);
  a2 = nmx_csub(t0, t2);
  a3 = nmx_add_ib<-DIR>(t1, d);
#else
  const nmx_c2 t3 = nmx_mul_i<DIR>(nmx_csub(a1, a3));
  a0 = nmx_cadd(t0, t2);
  a1 = nmx_cadd(t1, t3);
  a2 = nmx_csub(t0, t2);
  a3 = nmx_csub(t1, t3);
#endif
}

// in-register 16-point DFT: y_q = sum_r a_r w^(q r), w = exp(DIR 2 pi i / 16); in/out v[0..15]
template <int DIR>
NMX_DEV void nmx_dft16(nmx_c2* v) {
  // step 1: for each r0, DFT4 over r1 of a[r0 + 4 r1]  -> inner[r0][q1] stored at v[r0 + 4 q1]
NMX_UNROLL
  for (int r0 = 0; r0 < 4; ++r0) nmx_dft4<DIR>(v[r0], v[r0 + 4], v[r0 + 8], v[r0 + 12]);
  // step 2: twiddle inner[r0][q1] by w^(q1 r0)
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  const float sg = (float)DIR;
  // w^1 = (c1, sg s1), w^2 = (h, sg h), w^3 = (s1, sg c1), w^4 = (0, sg), w^6 = (-h, sg h), w^9 = (-c1, -sg s1)
  v[1 + 4] = nmx_cmul(v[1 + 4], nmx_mk2(c1, sg * s1));   // q1=1, r0=1 : w^1
  v[2 + 4] = nmx_cmul(v[2 + 4], nmx_mk2(h, sg * h));     // q1=1, r0=2 : w^2
  v[3 + 4] = nmx_cmul(v[3 + 4], nmx_mk2(s1, sg * c1));   // q1=1, r0=3 : w^3
  v[1 + 8] = nmx_cmul(v[1 + 8], nmx_mk2(h, sg * h));     // q1=2, r0=1 : w^2
  v[2 + 8] = nmx_mul_i<DIR>(v[2 + 8]);                       // q1=2, r0=2 : w^4
  v[3 + 8] = nmx_cmul(v[3 + 8], nmx_mk2(-h, sg * h));    // q1=2, r0=3 : w^6
  v[1 + 12] = nmx_cmul(v[1 + 12], nmx_mk2(s1, sg * c1));  // q1=3, r0=1 : w^3
  v[2 + 12] = nmx_cmul(v[2 + 12], nmx_mk2(-h, sg * h));   // q1=3, r0=2 : w^6
  v[3 + 12] = nmx_cmul(v[3 + 12], nmx_mk2(-c1, -sg * s1)); // q1=3, r0=3 : w^9
  // step 3: for each q1, DFT4 over r0 -> y[q1 + 4 q0] ; store so that v[q] = y_q
NMX_UNROLL
  for (int q1 = 0; q1 < 4; ++q1) nmx_dft4<DIR>(v[4 * q1], v[4 * q1 + 1], v[4 * q1 + 2], v[4 * q1 + 3]);
  // now v[4 q1 + q0] = y[q1 + 4 q0]: transpose the 4x4 index in registers
  nmx_c2 t;
#define NMX_SWAP(i, j) t = v[i]; v[i] = v[j]; v[j] = t;
  NMX_SWAP(1, 4) NMX_SWAP(2, 8) NMX_SWAP(3, 12) NMX_SWAP(6, 9) NMX_SWAP(7, 13) NMX_SWAP(11, 14)
#undef NMX_SWAP
}

// padded physical index of the pass-A output buffer
NMX_DEV int nmx_w64_pad(int idx) { return idx + (idx >> 4); }

// Passes A (from registers) .. C (to registers).  v: 16 points per lane, in: v[r] = x[lane + 64 r]
// out: v[4 t + r] = y[lane + 64 t + 256 r].  X = exchange buffer (>= 1024 + 64 nmx_c2).
// Phase functions are split so the host emulator can run them lane by lane.
template <int DIR>
NMX_DEV void nmx_w64_passA(nmx_c2* v, nmx_c2* X, int lane) {
  nmx_dft16<DIR>(v);
  nmx_c2* Xo = X + 17 * lane;  // == pad(16 lane + r) for r < 16: base + constant offsets
  NMX_UNROLL
  for (int r = 0; r < 16; ++r) Xo[r] = v[r];
}
NMX_DEV void nmx_w64_passB_store(const nmx_c2* v, nmx_c2* X, int lane) {
  nmx_c2* Xo = X + (lane >> 4) * 256 + (lane & 15);
NMX_UNROLL
  for (int r = 0; r < 16; ++r) Xo[16 * r] = v[r];
}
// Twiddles come from a table (LDS copy in the persistent kernels, L2 otherwise): exact values
// instead of products of a few base values, no twiddle registers, and none of the compile-time
// constants that v_pk_* instructions cannot take as literals (they spilled the scalar registers).
//   twB[(r - 1) * 16 + k] = exp(-2 pi i r k / 256),            r = 1..15, k = lane % 16
//   twC[((r - 1) * 4 + t) * 64 + lane] = exp(-2 pi i r (lane + 64 t) / 1024), r = 1..3, t = 0..3
#define NMX_W64_TWB_N (15 * 16)
#define NMX_W64_TWC_N (12 * 64)
#define NMX_W64_TWL_FLOATS (2 * (NMX_W64_TWB_N + NMX_W64_TWC_N))
#if defined(NMX_LDS_ASM) && !defined(NMX_HOST_EMU)
#include <utility>
// pass C order: v[4 t + q] = X[lane + 64 t + 256 q]
template <int... I>
NMX_DEV void nmx_ds_read_passC(nmx_c2* v, unsigned addr, std::integer_sequence<int, I...>) {
  ((v[I] = nmx_ds_read_b64<512 * (I / 4) + 2048 * (I % 4)>(addr)), ...);
}
// twC order: w[4 t + q] = twC[lane + 64 (4 (q - 1) + t)], q = 1..3 (slot 4 t unused)
template <int... I>
NMX_DEV void nmx_ds_read_twC(nmx_c2* w, unsigned addr, std::integer_sequence<int, I...>) {
  ((w[4 * (I / 3) + 1 + (I % 3)] = nmx_ds_read_b64<512 * (4 * (I % 3) + (I / 3))>(addr)), ...);
}
// TWLDS: the twiddle table is the LDS copy of the persistent / quad kernels (else global memory, L2)
template <int DIR, int TWLDS = 1>
NMX_DEV void nmx_w64_passB_load_lds(nmx_c2* v, const nmx_c2* X, const nmx_c2* twB, int lane) {
  nmx_c2 w[16];
  nmx_ds_read_seq<544, 0>(v, nmx_lds_addr(X + lane + (lane >> 4)), std::make_integer_sequence<int, 16>{});
  if (TWLDS) {
    nmx_ds_read_seq<128, 0>(w + 1, nmx_lds_addr(twB + (lane & 15)), std::make_integer_sequence<int, 15>{});
    w[0] = w[1];
    nmx_lds_wait8(v); nmx_lds_tie8(v + 8); nmx_lds_tie8(w); nmx_lds_tie8(w + 8);
  } else {
    const nmx_c2* tw = twB + (lane & 15);
    NMX_UNROLL
    for (int r = 1; r < 16; ++r) w[r] = tw[16 * (r - 1)];
    nmx_lds_wait8(v); nmx_lds_tie8(v + 8);
  }
  NMX_UNROLL
  for (int r = 1; r < 16; ++r) v[r] = nmx_cmul_tw<(DIR > 0)>(v[r], w[r]);
  nmx_dft16<DIR>(v);
}
#else
template <int DIR, int TWLDS = 1>
NMX_DEV void nmx_w64_passB_load_lds(nmx_c2* v, const nmx_c2* X, const nmx_c2* twB, int lane) {
  const nmx_c2* Xi = X + lane + (lane >> 4);
  NMX_UNROLL
  for (int r = 0; r < 16; ++r) v[r] = Xi[68 * r];
  const nmx_c2* tw = twB + (lane & 15);
  NMX_UNROLL
  for (int r = 1; r < 16; ++r) v[r] = nmx_cmul(v[r], nmx_twd<DIR>(tw[16 * (r - 1)]));
  nmx_dft16<DIR>(v);
}
#endif
#if defined(NMX_LDS_ASM) && !defined(NMX_HOST_EMU)
template <int DIR, int HALF, int TWLDS>
NMX_DEV void nmx_w64_passC_asm(nmx_c2* v, const nmx_c2* X, const nmx_c2* twC, int lane) {
  nmx_c2 a[16], w[16];
  nmx_ds_read_passC(a, nmx_lds_addr(X + lane), std::make_integer_sequence<int, 16>{});
  if (TWLDS) {
    nmx_ds_read_twC(w, nmx_lds_addr(twC + lane), std::make_integer_sequence<int, 12>{});
    w[0] = w[4] = w[8] = w[12] = w[1];
    nmx_lds_wait8(a); nmx_lds_tie8(a + 8); nmx_lds_tie8(w); nmx_lds_tie8(w + 8);
  } else {
    const nmx_c2* tw = twC + lane;
    NMX_UNROLL
    for (int t = 0; t < 4; ++t) { w[4 * t + 1] = tw[64 * t]; w[4 * t + 2] = tw[64 * (4 + t)]; w[4 * t + 3] = tw[64 * (8 + t)]; }
    nmx_lds_wait8(a); nmx_lds_tie8(a + 8);
  }
NMX_UNROLL
  for (int t = 0; t < 4; ++t) {
    nmx_c2 a0 = a[4 * t], a1 = a[4 * t + 1], a2 = a[4 * t + 2], a3 = a[4 * t + 3];
    a1 = nmx_cmul_tw<(DIR > 0)>(a1, w[4 * t + 1]);
    a2 = nmx_cmul_tw<(DIR > 0)>(a2, w[4 * t + 2]);
    a3 = nmx_cmul_tw<(DIR > 0)>(a3, w[4 * t + 3]);
    if (HALF) {
      const nmx_c2 t0 = nmx_cadd(a0, a2), t1 = nmx_csub(a0, a2), t2 = nmx_cadd(a1, a3);
      v[4 * t] = nmx_cadd(t0, t2);
      v[4 * t + 1] = nmx_add_ib<DIR>(t1, nmx_csub(a1, a3));
    } else {
      nmx_dft4<DIR>(a0, a1, a2, a3);
      v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
    }
  }
}
template <int DIR, int TWLDS = 1>
NMX_DEV void nmx_w64_passC_lds(nmx_c2* v, const nmx_c2* X, const nmx_c2* twC, int lane) {
  nmx_w64_passC_asm<DIR, 0, TWLDS>(v, X, twC, lane);
}
template <int DIR, int TWLDS = 1>
NMX_DEV void nmx_w64_passC_lds_half(nmx_c2* v, const nmx_c2* X, const nmx_c2* twC, int lane) {
  nmx_w64_passC_asm<DIR, 1, TWLDS>(v, X, twC, lane);
}
#define NMX_W64_PASSC_DEFINED 1
#endif
#ifndef NMX_W64_PASSC_DEFINED
template <int DIR, int TWLDS = 1>
NMX_DEV void nmx_w64_passC_lds(nmx_c2* v, const nmx_c2* X, const nmx_c2* twC, int lane) {
  const nmx_c2* tw = twC + lane;
NMX_UNROLL
  for (int t = 0; t < 4; ++t) {
    const nmx_c2* Xi = X + lane;
    nmx_c2 a0 = Xi[64 * t], a1 = Xi[64 * t + 256], a2 = Xi[64 * t + 512], a3 = Xi[64 * t + 768];
    a1 = nmx_cmul(a1, nmx_twd<DIR>(tw[64 * t]));
    a2 = nmx_cmul(a2, nmx_twd<DIR>(tw[64 * (4 + t)]));
    a3 = nmx_cmul(a3, nmx_twd<DIR>(tw[64 * (8 + t)]));
    nmx_dft4<DIR>(a0, a1, a2, a3);
    v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
  }
}

// Inverse pass C when only the first half of the outputs is read (W <= 1024 samples = packed index
// m < 512 = the r = 0, 1 outputs of each 4-point butterfly): two of the four outputs, 6 instead of 8 adds.
template <int DIR, int TWLDS = 1>
NMX_DEV void nmx_w64_passC_lds_half(nmx_c2* v, const nmx_c2* X, const nmx_c2* twC, int lane) {
  const nmx_c2* tw = twC + lane;
NMX_UNROLL
  for (int t = 0; t < 4; ++t) {
    const nmx_c2* Xi = X + lane;
    nmx_c2 a0 = Xi[64 * t], a1 = Xi[64 * t + 256], a2 = Xi[64 * t + 512], a3 = Xi[64 * t + 768];
    a1 = nmx_cmul(a1, nmx_twd<DIR>(tw[64 * t]));
    a2 = nmx_cmul(a2, nmx_twd<DIR>(tw[64 * (4 + t)]));
    a3 = nmx_cmul(a3, nmx_twd<DIR>(tw[64 * (8 + t)]));
    const nmx_c2 t0 = nmx_cadd(a0, a2), t1 = nmx_csub(a0, a2), t2 = nmx_cadd(a1, a3);
    const nmx_c2 t3 = nmx_mul_i<DIR>(nmx_csub(a1, a3));
    v[4 * t] = nmx_cadd(t0, t2);
    v[4 * t + 1] = nmx_cadd(t1, t3);
  }
}
#endif

// tail-range mask of a register that straddles the band-pass segment boundary: kept out of line so
// that the compiler does not if-convert it into 16 x 2 lane predicates (64-bit masks that were
// spilling the scalar register file); it runs for at most two registers per pass
#ifdef NMX_HOST_EMU
#define NMX_NOINLINE static
#else
#define NMX_NOINLINE __device__ __attribute__((noinline))
#endif
NMX_NOINLINE nmx_c2 nmx_w64_range_mask(nmx_c2 val, int m, int lo, int hi) {
  return nmx_mk2((2 * m >= lo && 2 * m < hi) ? val.x : 0.f, (2 * m + 1 >= lo && 2 * m + 1 < hi) ? val.y : 0.f);
}

// wave reductions (single-wave workgroup)
#ifdef NMX_HOST_EMU
#define NMX_W64_REDUCE_SUM(acc_array, result)      \
  { float s_ = 0.f; for (int l = 0; l < 64; ++l) s_ += acc_array[l]; result = s_; }
#else
#define NMX_W64_REDUCE_SUM(acc_array, result) \
  { result = nmx_wave_reduce(acc_array[0], 0.f, [](float a_, float b_) { return a_ + b_; }); }
#endif

#ifndef NMX_HOST_EMU
// the notch's reflection table (see nmx_bank_w64_item, rtab): [16][64] words, built by the threads of a workgroup
NMX_DEV void nmx_w64_reflect_table(const NmxBankArgs& A, unsigned* rtab, int tid, int nthreads) {
  const int W = A.W, h = A.pad_half, ne = A.n_edge;
  for (int i = tid; i < 1024; i += nthreads) {
    const int r = i >> 6, l = i & 63;
    unsigned word = 0;
    for (int u = 0; u < 2; ++u) {
      const int jp = 2 * (l + 64 * r) + u, j = jp - h;
      unsigned e = 0xfffcu;   // beyond the staged signal / the reflection limit: loads 0, enters as is
      if (jp < W + 2 * h) {
        if (j < 0) { if (-j <= ne) e = (unsigned)(4 * -j) | 1u; }
        else if (j < W) e = (unsigned)(4 * j);
        else { const int rr = j - (W - 1); if (rr <= ne) e = (unsigned)(4 * (W - 1 - rr)) | 2u; }
      }
      word |= e << (16 * u);
    }
    rtab[i] = word;
  }
}
#endif

// PAD = 0: zero-padded window ("same" FIR bank);  PAD = 1: odd-reflected window (notch)
// TAB = 1: the A/B tables of all filters sit in LDS at `tab` ([filter][A[n], B[n]]), staged once
// per (persistent, multi-wave) workgroup; TAB = 0: read from global memory (L2).
// HALF = 1 (PAD = 0, W <= 1024): only the output registers v[4 t + r], r < 2 (samples < 1024) exist -- the
// others are never computed, reduced or stored.
template <int PAD, int TAB, int MC, int HALF = 0, int HOIST = 1>
NMX_DEV void nmx_bank_w64_item(const NmxBankW64Args& AA, int w, int c, float* smem, const float* tab,
                               const unsigned* rtab = nullptr) {
  w = nmx_uniform_i(w);   // one item per wave: (w, c) and everything derived from them is scalar
  c = nmx_uniform_i(c);
  const NmxBankArgs& A = AA.b;
  nmx_c2* X = (nmx_c2*)(smem + AA.off_X);   // [1024 + 64] exchange buffer (the only LDS tile)
  float* red = smem + AA.off_red;
  const int W = A.W, n = NMX_W64_N;
  float* out_row = A.out ? A.out + (long long)w * A.n_outputs : nullptr;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  nmx_c2 v[NMX_LANES][16];
  nmx_c2 zr[NMX_LANES][16];   // forward transform Z, kept in registers: zr[4 t + r] = Z[l + 64 t + 256 r]
  // twiddle table: LDS copy in the persistent kernels, the same values from global memory (L2)
  // otherwise -- every variant computes bit-identical results
  const nmx_c2* twB = TAB ? (const nmx_c2*)(tab + (size_t)A.n_filters * 2 * NMX_W64_N) : (const nmx_c2*)AA.twl;
  const nmx_c2* twC = twB + NMX_W64_TWB_N;
  NMX_PROF_DECL

  // ---- forward: window -> registers (packed complex, lane-consecutive) -> pass A ------------
#ifdef NMX_HOST_EMU
  if (PAD == 1) {  // notch: stage the window in LDS for the odd reflection
    float* xs = (float*)X;
    NMX_LANE_LOOP {
      // 16 loads in flight before the first LDS store; a second batch for windows beyond 1024 samples
      // (W + 2 * pad_half <= 2048 on this path, and the exchange buffer holds 2176 floats)
      for (int base = 0; base < W; base += 1024) {
        float t[16];
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int i = base + l + 64 * r;
          t[r] = i < W ? src[i] : 0.f;
        }
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          const int i = base + l + 64 * r;
          if (i < W) xs[i] = A.clean_on_load ? nmx_clean(t[r]) : t[r];
        }
      }
    }
    NMX_WSYNC();
  }
#endif
  NMX_LANE_LOOP {
    nmx_c2* vv = v[NMX_LI];
    if (PAD == 0) {
#ifdef NMX_HOST_EMU
      NMX_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int n0 = 2 * (l + 64 * r);
        float v0 = n0 < W ? src[n0] : 0.f, v1 = (n0 + 1) < W ? src[n0 + 1] : 0.f;
        if (A.clean_on_load) { v0 = nmx_clean(v0); v1 = nmx_clean(v1); }
        vv[r] = nmx_mk2(v0, v1);
      }
#else
      const nmx_rsrc rs = nmx_make_rsrc(src, 4 * W);
      if ((W & 1) == 0) {
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) vv[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, 8 * l + 512 * r, 0, 0);
      } else {   // a pair may straddle the end of the row: dword accesses are range-checked one by one
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          vv[r].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 8 * l + 512 * r, 0, 0));
          vv[r].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 8 * l + 512 * r + 4, 0, 0));
        }
      }
      if (A.clean_on_load) {
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) vv[r] = nmx_mk2(nmx_clean_bl(vv[r].x), nmx_clean_bl(vv[r].y));
      }
#endif
    } else {
#ifndef NMX_HOST_EMU
      // Notch: the odd-reflected window straight from global memory (L2), no LDS staging.  Sample jp of the staged
      // signal is x[j], j = jp - h, inside the window and  2 x[0] - x[-j]  /  2 x[W-1] - x[2 (W-1) - j]  in the
      // reflected flanks (MNE _smart_pad, reflect_limited: at most n_edge samples); every lane computes its 32
      // indices branch-free and all loads are in flight before the first use (the staged form walked 18 of the 32
      // elements through nested branches with one dependent LDS read each: 60 % of the notch item's cycles).
      if (rtab) {
        // The persistent kernel's form: WHERE sample jp of the staged signal comes from and HOW it enters (as is, or
        // reflected about the first / the last sample) depends on the plan and the lane only -- nmx_w64_reflect_table
        // writes it once per workgroup into LDS, two 16-bit entries per word: byte offset | code (0: x, 1: 2 x[0] - x,
        // 2: 2 x[W-1] - x; a sample beyond the staged signal is code 0 at an out-of-range offset, which loads 0).  Per
        // sample: decode + load + one subtraction and two selects instead of ~20 index / range instructions.
        const nmx_rsrc rs = nmx_make_rsrc(src, 4 * W);
        unsigned wds[16];
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) wds[r] = rtab[64 * r + l];
        float t[32];
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          t[2 * r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(wds[r] & 0xfffcu), 0, 0));
          t[2 * r + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((wds[r] >> 16) & 0xfffcu), 0, 0));
        }
        float x0 = src[0], xl = src[W - 1];
        if (A.clean_on_load) { x0 = nmx_clean(x0); xl = nmx_clean(xl); }
        const float x02 = 2.f * x0, xl2 = 2.f * xl;
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          float e2[2];
          NMX_UNROLL
          for (int u = 0; u < 2; ++u) {
            const unsigned code = (wds[r] >> (16 * u)) & 3u;
            float x = t[2 * r + u];
            if (A.clean_on_load) x = nmx_clean_bl(x);
            const float refl = (code == 1u ? x02 : xl2) - x;
            e2[u] = code == 0u ? x : refl;
          }
          vv[r] = nmx_mk2(e2[0], e2[1]);
        }
      } else
      {
        const nmx_rsrc rs = nmx_make_rsrc(src, 4 * W);
        const int h = A.pad_half, ne = A.n_edge;
        // (branch-free for every register: wave-uniform region tests -- interior / left flank / right flank / beyond the
        // staged signal, one subtraction each -- execute a third of the instructions and were SLOWER, 1.12 -> 1.30 ms:
        // the branches between the load groups cost more than the index arithmetic they save)
        float t[32];
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          NMX_UNROLL
          for (int u = 0; u < 2; ++u) {
            const int j = 2 * (l + 64 * r) + u - h;
            const int idx = j < 0 ? -j : (j >= W ? 2 * (W - 1) - j : j);
            t[2 * r + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * idx, 0, 0));   // out of range: 0
          }
        }
        float x0 = src[0], xl = src[W - 1];
        if (A.clean_on_load) { x0 = nmx_clean(x0); xl = nmx_clean(xl); }
        NMX_UNROLL
        for (int r = 0; r < 16; ++r) {
          float e2[2];
          NMX_UNROLL
          for (int u = 0; u < 2; ++u) {
            const int jp = 2 * (l + 64 * r) + u, j = jp - h;
            float x = t[2 * r + u];
            if (A.clean_on_load) x = nmx_clean_bl(x);
            const bool lo = j < 0, hi = j >= W;
            const int dist = lo ? -j : j - (W - 1);
            const float refl = (lo ? 2.f * x0 : 2.f * xl) - x;
            const float val = (lo || hi) ? (dist <= ne ? refl : 0.f) : x;
            e2[u] = jp < W + 2 * h ? val : 0.f;
          }
          vv[r] = nmx_mk2(e2[0], e2[1]);
        }
      }
#else
      const float* xs = (const float*)X;
      const int h = A.pad_half, ne = A.n_edge;
      const float x0 = xs[0], xl = xs[W - 1];
      NMX_UNROLL
      for (int r = 0; r < 16; ++r) {
        if (128 * r >= h && 128 * r + 127 < W + h) {   // wave-uniform: this register is all interior
          const float* q = xs + (2 * l + 128 * r - h);
          vv[r] = nmx_mk2(q[0], q[1]);
          continue;
        }
        float e2[2];
        for (int u = 0; u < 2; ++u) {
          const int jp = 2 * (l + 64 * r) + u;
          float val = 0.f;
          if (jp < W + 2 * h) {
            const int j = jp - h;
            if (j < 0) val = (-j <= ne) ? 2.f * x0 - xs[-j] : 0.f;
            else if (j < W) val = xs[j];
            else { const int rr = j - (W - 1); val = (rr <= ne) ? 2.f * xl - xs[W - 1 - rr] : 0.f; }
          }
          e2[u] = val;
        }
        vv[r] = nmx_mk2(e2[0], e2[1]);
      }
#endif
    }
  }
  NMX_WSYNC();  // (emulator, pad_mode 1: everyone has read xs before X is overwritten)
  NMX_LANE_LOOP { nmx_w64_passA<-1>(v[NMX_LI], X, l); }
  NMX_WSYNC();
  NMX_LANE_LOOP { nmx_w64_passB_load_lds<-1, TAB>(v[NMX_LI], X, twB, l); }
  NMX_WSYNC();
  NMX_LANE_LOOP { nmx_w64_passB_store(v[NMX_LI], X, l); }
  NMX_WSYNC();
  NMX_LANE_LOOP {
    nmx_c2* vv = v[NMX_LI];
    nmx_w64_passC_lds<-1, TAB>(vv, X, twC, l);
    NMX_UNROLL
    for (int i = 0; i < 16; ++i) zr[NMX_LI][i] = vv[i];
  }
  NMX_WSYNC();
  NMX_PROF(0)

#ifndef NMX_HOST_EMU
  // conjugate partners Z[n - k] of this lane's points: one cross-lane read per point, ONCE per item (they
  // do not depend on the filter; inside the filter loop they were a third of its LDS-crossbar traffic)
  nmx_c2 zcr[16];
  if (PAD == 0 && HOIST) {   // (the notch has one filter: nothing to hoist, and it needs its 3 waves/SIMD)
    const int l = (int)(threadIdx.x & 63);
    NMX_UNROLL
    for (int r = 0; r < 16; ++r) {
      const nmx_c2 zs = zr[0][NMX_J2I(15 - r)];
      nmx_c2 zc = nmx_mk2(__shfl(zs.x, (64 - l) & 63), __shfl(zs.y, (64 - l) & 63));
      if (l == 0) zc = (r == 0) ? zr[0][0] : zr[0][NMX_J2I((16 - r) & 15)];
      zcr[r] = zc;
    }
  }
#endif
  NMX_PROF(6)
  const int yoff = (PAD == 1) ? A.pad_half : 0;
  for (int fi = 0; fi < A.n_filters; ++fi) {
    const NmxFilterDev& F = A.f[fi];
    const float* NMX_RESTRICT Hs = TAB ? tab + (size_t)fi * 2 * NMX_W64_N : AA.Hs[fi];
    const float* NMX_RESTRICT Hd = TAB ? tab + (size_t)fi * 2 * NMX_W64_N + NMX_W64_N : AA.Hd[fi];
    // ---- fused split * H * unsplit into registers, then inverse passes -----------------------
    NMX_LANE_LOOP {
      nmx_c2* vv = v[NMX_LI];
      NMX_UNROLL
      for (int r = 0; r < 16; ++r) {
        // k = l + 64 r lives in register j2i(r); its partner n - k = (64 - l) + 64 (15 - r) lives in
        // lane (64 - l) & 63, register j2i(15 - r)  (lane 0: its own register j2i(16 - r), Z[n] = Z[0])
        const nmx_c2 zk = zr[NMX_LI][NMX_J2I(r)];
#ifdef NMX_HOST_EMU
        nmx_c2 zc = zr[(64 - l) & 63][NMX_J2I(15 - r)];
        if (l == 0) zc = (r == 0) ? zr[NMX_LI][0] : zr[NMX_LI][NMX_J2I((16 - r) & 15)];
#else
        nmx_c2 zc;
        if (PAD == 0 && HOIST) {
          zc = zcr[r];
        } else {
          const nmx_c2 zs = zr[0][NMX_J2I(15 - r)];
          zc = nmx_mk2(__shfl(zs.x, (64 - l) & 63), __shfl(zs.y, (64 - l) & 63));
          if (l == 0) zc = (r == 0) ? zr[0][0] : zr[0][NMX_J2I((16 - r) & 15)];
        }
#endif
        const float ha = (Hs + l)[64 * r], hb = (Hd + l)[64 * r];
        // Z'[k] = A_k Z[k] + i B_k conj(Z[n-k]),  A = Hs - Hd sin(th_k), B = Hd cos(th_k)
        vv[r] = nmx_c2_axpby_swap(ha, zk, hb, zc);
        if ((r & 3) == 3) NMX_SCHED_FENCE();
      }
      nmx_w64_passA<+1>(vv, X, l);
    }
    NMX_WSYNC();
    NMX_PROF(1)
    NMX_LANE_LOOP { nmx_w64_passB_load_lds<+1, TAB>(v[NMX_LI], X, twB, l); }
    NMX_WSYNC();
    NMX_PROF(2)
    NMX_LANE_LOOP { nmx_w64_passB_store(v[NMX_LI], X, l); }
    NMX_WSYNC();
    NMX_PROF(3)
    NMX_LANE_LOOP {
      if (HALF) nmx_w64_passC_lds_half<+1, TAB>(v[NMX_LI], X, twC, l);
      else nmx_w64_passC_lds<+1, TAB>(v[NMX_LI], X, twC, l);
    }
    NMX_PROF(4)
    // now lane l holds y[2 m], y[2 m + 1] in v[4 t + r] for m = l + 64 t + 256 r

    if (F.bp_seglen > 0) {
      const bool need_mc = MC && (A.bp_features & 6u) != 0;   // MC = 0: compiled for activity only
      if (!need_mc) {  // activity only: variance straight from registers
        // Register i of every lane covers samples [2 mb, 2 mb + 127]: all but the (at most two)
        // registers that straddle lo or hi are wave-uniformly inside or outside the tail, so the
        // range test is a scalar branch and the sums are packed adds / fmas.
        // ONE pass: sum and sum of squares together, var = E[y^2] - E[y]^2.  The series is the output
        // of a band-pass filter, so E[y]^2 << E[y^2] and the subtraction costs no accuracy in fp32
        // (a second, mean-shifted pass over the registers was 8 % of the kernel).
        const int lo = W - F.bp_seglen + yoff, hi = W + yoff;
        float part[NMX_LANES], part2[NMX_LANES];
        NMX_LANE_LOOP {
          nmx_c2 acc = nmx_mk2(0.f, 0.f), acc2 = nmx_mk2(0.f, 0.f);
#if defined(NMX_HOST_EMU)
          NMX_UNROLL
          for (int i = 0; i < 16; ++i) {
            if (HALF && (i & 3) >= 2) continue;
            const int mb = 64 * (i >> 2) + 256 * (i & 3);
            if (2 * mb + 127 < lo || 2 * mb >= hi) continue;
            nmx_c2 val = v[NMX_LI][i];
            if (!(2 * mb >= lo && 2 * mb + 127 < hi)) val = nmx_w64_range_mask(val, l + mb, lo, hi);
            acc = nmx_cadd(acc, val);
            acc2 = nmx_c2_fma(val, val, acc2);
          }
#else
          // BRANCH-FREE: every register is weighed per lane by [lo <= sample < hi] (one unsigned compare per
          // sample) and always accumulated -- adding the zeros of the registers outside the tail changes no
          // bit.  The wave-uniform skip / straddle tests this replaces were ~20 scalar branches and up to three
          // out-of-line calls per filter: measured (s_memtime, -DNMX_BANK_PROFILE) at ~2 400 cycles per filter,
          // a quarter of the whole item, for 16 packed operations of arithmetic.
          const unsigned span = (unsigned)(hi - lo);
          const int s_l = 2 * l - lo;
          NMX_UNROLL
          for (int i = 0; i < 16; ++i) {
            if (HALF && (i & 3) >= 2) continue;
            const int sb = s_l + 2 * (64 * (i >> 2) + 256 * (i & 3));
            nmx_c2 val = v[NMX_LI][i];
            val.x = (unsigned)sb < span ? val.x : 0.f;
            val.y = (unsigned)(sb + 1) < span ? val.y : 0.f;
            acc = nmx_cadd(acc, val);
            acc2 = nmx_c2_fma(val, val, acc2);
          }
#endif
          part[NMX_LI] = acc.x + acc.y;
          part2[NMX_LI] = acc2.x + acc2.y;
        }
        float tot, tot2;
        NMX_W64_REDUCE_SUM(part, tot);
        NMX_W64_REDUCE_SUM(part2, tot2);
        const float mean = tot / (float)F.bp_seglen;
        tot = tot2 - mean * tot;   // sum (y - mean)^2 = sum y^2 - mean sum y
        if (mean * mean * (float)F.bp_seglen > 4.f * tot) {
          // wave-uniform, rare (a short tail of a slow band is almost a constant): the subtraction
          // above cancels, redo it mean-shifted like np.var
          NMX_LANE_LOOP {
            nmx_c2 acc = nmx_mk2(0.f, 0.f);
            const nmx_c2 mean2 = nmx_mk2(mean, mean);
            NMX_UNROLL
            for (int i = 0; i < 16; ++i) {
              if (HALF && (i & 3) >= 2) continue;
              const int mb = 64 * (i >> 2) + 256 * (i & 3);
              if (2 * mb + 127 < lo || 2 * mb >= hi) continue;
              nmx_c2 d = nmx_csub(v[NMX_LI][i], mean2);
              if (!(2 * mb >= lo && 2 * mb + 127 < hi)) d = nmx_w64_range_mask(d, l + mb, lo, hi);
              acc = nmx_c2_fma(d, d, acc);
            }
            part[NMX_LI] = acc.x + acc.y;
          }
          NMX_W64_REDUCE_SUM(part, tot);
        }
        const float act = tot / (float)F.bp_seglen;
        NMX_LANE_LOOP {
          if (l == 0) {
            const int col = A.bp_cols.base + c * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
            out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u);
          }
        }
      } else {  // mobility / complexity need neighbours: go through LDS (natural order)
        NMX_WSYNC();
        NMX_LANE_LOOP {
          NMX_UNROLL
          for (int i = 0; i < 16; ++i) X[l + 64 * (i >> 2) + 256 * (i & 3)] = v[NMX_LI][i];
        }
        NMX_WSYNC();
        const float* y = (const float*)X + yoff;
        float act, mob, comp;
        nmx_hjorth(y + (W - F.bp_seglen), F.bp_seglen, red, 1, true, act, mob, comp);
        if (NMX_TID == 0) {
          int col = A.bp_cols.base + c * A.bp_cols.ch_stride + F.bp_band * A.bp_cols.a_stride;
          if (A.bp_features & 1u) { out_row[col] = nmx_bp_activity(A.bp_log ? log10f(act) : act, (A.bp_kalman_mask >> F.bp_band) & 1u); col += A.bp_cols.b_stride; }
          if (A.bp_features & 2u) { out_row[col] = nmx_nan_to_num(mob); col += A.bp_cols.b_stride; }
          if (A.bp_features & 4u) out_row[col] = nmx_nan_to_num(comp);
        }
      }
    }
    NMX_PROF(5)
    // ---- filtered series to HBM (lane-consecutive) ---------------------------------------------
    if (PAD == 0) {
      float* dsw = F.sw_index >= 0
          ? A.sw_out + (((long long)w * A.n_channels + c) * A.n_sw_filters + F.sw_index) * W : nullptr;
      float* dyb = F.burst_index >= 0
          ? AA.yb_out + (((long long)w * A.n_channels + c) * A.n_burst_bands + F.burst_index) * W : nullptr;
#ifdef NMX_HOST_EMU
      NMX_LANE_LOOP {
        for (int i = 0; i < 16; ++i) {
          const int m = l + 64 * (i >> 2) + 256 * (i & 3);
          const nmx_c2 val = v[NMX_LI][i];
          if (2 * m < W) { if (dsw) dsw[2 * m] = val.x; if (dyb) dyb[2 * m] = val.x; }
          if (2 * m + 1 < W) { if (dsw) dsw[2 * m + 1] = val.y; if (dyb) dyb[2 * m + 1] = val.y; }
        }
      }
#else
      // wave-uniform choice of destinations; the ragged row end is the buffer range check
      const int l = (int)(threadIdx.x & 63);
      for (int dst = 0; dst < 2; ++dst) {
        float* d = dst ? dyb : dsw;
        if (!d) continue;
        const nmx_rsrc rs = nmx_make_rsrc(d, 4 * W);
        if ((W & 1) == 0) {
          NMX_UNROLL
          for (int i = 0; i < 16; ++i) {
            if (HALF && (i & 3) >= 2) continue;
            __builtin_amdgcn_raw_buffer_store_b64(v[0][i], rs, 8 * l + 512 * (i >> 2) + 2048 * (i & 3), 0, NMX_SERIES_STORE_AUX);
          }
        } else {
          NMX_UNROLL
          for (int i = 0; i < 16; ++i) {
            if (HALF && (i & 3) >= 2) continue;
            const int off = 8 * l + 512 * (i >> 2) + 2048 * (i & 3);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0][i].x), rs, off, 0, NMX_SERIES_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0][i].y), rs, off + 4, 0, NMX_SERIES_STORE_AUX);
          }
        }
      }
#endif
    } else if (F.store_raw) {
#ifndef NMX_HOST_EMU
      {   // offsets relative to sample yoff: what lies in front wraps to a huge unsigned offset, what lies behind
          // exceeds num_records -- both are dropped by the buffer unit, no compares
        const int l = (int)(threadIdx.x & 63);
        const nmx_rsrc rd = nmx_make_rsrc(A.y_out + ((long long)w * A.n_channels + c) * W, 4 * W);
        NMX_UNROLL
        for (int i = 0; i < 16; ++i) {
          const int s0 = 2 * (l + 64 * (i >> 2) + 256 * (i & 3)) - yoff;
          nmx_c2 val = v[0][i];
          if (A.residual) {   // x - g * x_ext (NmxBankArgs::residual): the window's samples again, from L2
            const nmx_rsrc rx = nmx_make_rsrc(src, 4 * W);
            nmx_c2 xw = nmx_mk2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, 4 * s0, 0, 0)),
                                __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, 4 * s0 + 4, 0, 0)));
            if (A.clean_on_load) xw = nmx_mk2(nmx_clean_bl(xw.x), nmx_clean_bl(xw.y));
            val = xw - val;
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val.x), rd, 4 * s0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val.y), rd, 4 * s0 + 4, 0, 0);
        }
      }
#else
      float* d2 = A.y_out + ((long long)w * A.n_channels + c) * W - yoff;
      NMX_LANE_LOOP {
        NMX_UNROLL
        for (int i = 0; i < 16; ++i) {
          const int s0 = 2 * (l + 64 * (i >> 2) + 256 * (i & 3));
          const nmx_c2 val = v[NMX_LI][i];
          if (s0 >= yoff && s0 < W + yoff)
            d2[s0] = A.residual ? (A.clean_on_load ? nmx_clean(src[s0 - yoff]) : src[s0 - yoff]) - val.x : val.x;
          if (s0 + 1 >= yoff && s0 + 1 < W + yoff)
            d2[s0 + 1] = A.residual ? (A.clean_on_load ? nmx_clean(src[s0 + 1 - yoff]) : src[s0 + 1 - yoff]) - val.y : val.y;
        }
      }
#endif
    }
    NMX_WSYNC();
    NMX_PROF(7)
  }
  NMX_PROF_PRINT(w, c)
}

// ---- Hilbert envelope kernel: y[item][W] -> |analytic(y)| (exact length-W transforms) ----------
struct NmxHilbertArgs {
  const float* y;   // [n_items][W]
  float* env;       // [n_items][W]
  int W;
  NmxFft hil_r;     // complex length W/2 (W even) or W (odd)
  NmxFft hil_c;     // complex length W
  int hil_full;
  int off_a, off_b, off_y, lds_floats;
  const float* w500_tab;   // W = 1000: tables of the wave-level kernel (nmx_k_fft500.h)
  const float* w1000_tab;  // W = 2000: tables of the 1000-point wave-level transform (nmx_k_fft500.h)
};

NMX_DEV void nmx_hilbert_item(const NmxHilbertArgs& A, long long item, float* smem) {
  float2* bufA = (float2*)(smem + A.off_a);
  float2* bufB = (float2*)(smem + A.off_b);
  const int W = A.W, Wh = W >> 1;
  const float* src = A.y + item * W;
  float* dst = A.env + item * W;
  if (!A.hil_full) {
    // Even W.  The analytic signal of a real series is y + i H[y]; its real part IS y, and H[y] is real:
    // H[y] = irfft(Z), Z[k] = -i Y[k] (0 < k < W/2), Z[0] = Z[W/2] = 0 (scipy.signal.hilbert weights
    // the DC and Nyquist bins by 1: they only reach the real part).  So ONE half-length complex
    // transform each way instead of a half-length forward plus a full-length complex inverse.
    float* ys = smem + A.off_y;
    float* pk = (float*)bufB;   // packed complex: (x[2i], x[2i+1])
    nmx_stage_row(src, W, [=](int i, float v) { pk[i] = v; ys[i] = v; });
    NMX_SYNC();
    const float2* Zy = nmx_fft_auto<-1, true>(A.hil_r, bufB, bufA, bufB);
    float2* Yb = (Zy == bufA) ? bufB : bufA;     // Hermitian half Y[0 .. W/2]
    for (int k = NMX_TID; k <= Wh; k += NMX_NT) Yb[k] = nmx_rfft_bin(Zy, A.hil_r.twr, Wh, k);
    NMX_SYNC();
    float2* Pre = (Yb == bufA) ? bufB : bufA;    // overwrites Zy
    for (int k = NMX_TID; k < Wh; k += NMX_NT) {
      const float2 yk = Yb[k], yn = Yb[Wh - k];
      const float2 xk = k == 0 ? make_float2(0.f, 0.f) : make_float2(yk.y, -yk.x);    // -i Y[k]
      const float2 xn = k == 0 ? make_float2(0.f, 0.f) : make_float2(yn.y, -yn.x);    // -i Y[W/2 - k]
      Pre[k] = nmx_irfft_pre(xk, xn, A.hil_r.twr[k]);
    }
    NMX_SYNC();
    const float* ht = (const float*)nmx_fft_auto<+1, true>(A.hil_r, Pre, Yb, Pre);   // W * H[y], natural order
    const float invW = 1.f / (float)W;
    for (int i = NMX_TID; i < W; i += NMX_NT) {
      const float re = ys[i], im = ht[i] * invW;
      dst[i] = nmx_sqrt_fast(re * re + im * im);
    }
    return;
  }
  // odd W: full-length complex transforms
  nmx_stage_row(src, W, [=](int i, float v) { bufB[i] = make_float2(v, 0.f); });
  NMX_SYNC();
  const float2* Zy = nmx_fft_auto<-1, true>(A.hil_r, bufB, bufA, bufB);
  float2* Ab = (Zy == bufA) ? bufB : bufA;
  const float invW = 1.f / (float)W;
  for (int k = NMX_TID; k < W; k += NMX_NT) {
    float2 val = make_float2(0.f, 0.f);
    if (k == 0) val = Zy[0];
    else if (k <= (W - 1) / 2) val = make_float2(2.f * Zy[k].x, 2.f * Zy[k].y);
    Ab[k] = make_float2(val.x * invW, val.y * invW);
  }
  NMX_SYNC();
  float2* Zbuf = (Ab == bufA) ? bufB : bufA;
  const float2* an = nmx_fft_auto<+1, true>(A.hil_c, Ab, Zbuf, Ab);
  for (int i = NMX_TID; i < W; i += NMX_NT) dst[i] = nmx_sqrt_fast(an[i].x * an[i].x + an[i].y * an[i].y);
}

#ifndef NMX_HOST_EMU
// Hilbert envelope of one length-1000 series per WAVE (same arithmetic as the fused tail of the
// persistent bank kernel: both call nmx_w500_hilbert with the same tables).
// LDS per wave: a[500] + b[501] complex.
#define NMX_W500_LDS_FLOATS (1008 + 1000)
NMX_DEV void nmx_hilbert_w500_item(const NmxHilbertArgs& A, long long item, float* smem) {
  const int l = NMX_TID;
  nmx_c2* hb = (nmx_c2*)smem;
  nmx_c2* ha = hb + 504;
#ifdef NMX_DEBUG_HILBERT_SAMEROW   // (bound experiment, wrong results: every series comes from 4 MB that stay in L2)
  const nmx_rsrc rin = nmx_make_rsrc(A.y + (item & 1023) * 1000, 4000);
#else
  const nmx_rsrc rin = nmx_make_rsrc(A.y + item * 1000, 4000);
#endif
  const nmx_rsrc rout = nmx_make_rsrc(A.env + item * 1000, 4000);
  NmxW500TwReg T;
  T.load(A.w500_tab, l);
  nmx_c2 y[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) y[q] = __builtin_amdgcn_raw_buffer_load_b64(rin, 8 * l + 512 * q, 0, 0);
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q < 7 || l < 52) hb[l + 64 * q] = y[q];
  NMX_WAVE_FENCE();
  const nmx_c2* ht = nmx_w500_hilbert(ha, hb, T, (const nmx_c2*)A.w500_tab + NMX_W500_TW_N, l);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const nmx_c2 h = ht[l + 64 * q];
    const nmx_c2 e = nmx_mk2(nmx_sqrt_fast(y[q].x * y[q].x + h.x * h.x), nmx_sqrt_fast(y[q].y * y[q].y + h.y * h.y));
    __builtin_amdgcn_raw_buffer_store_b64(e, rout, 8 * l + 512 * q, 0, 0);
  }
}
// The same for length-2000 series (BASELINE config 3): 1000 packed points, 1000-point wave-level transforms.
// LDS per wave: ONE buffer of 1000 complex points (the transforms run in place) = 8 KB.
#define NMX_W1000_LDS_FLOATS 2000
NMX_DEV void nmx_hilbert_w1000_item(const NmxHilbertArgs& A, long long item, float* smem) {
  const int l = NMX_TID;
  nmx_c2* hb = (nmx_c2*)smem;
  nmx_c2* ha = hb;
  const nmx_rsrc rin = nmx_make_rsrc(A.y + item * 2000, 8000);
  const nmx_rsrc rout = nmx_make_rsrc(A.env + item * 2000, 8000);
  NmxW1000TwReg T;
  T.load(A.w1000_tab, l);
  nmx_c2 y[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) y[q] = __builtin_amdgcn_raw_buffer_load_b64(rin, 8 * l + 512 * q, 0, 0);
#pragma unroll
  for (int q = 0; q < 16; ++q)
    if (q < 15 || l < 40) hb[l + 64 * q] = y[q];
  NMX_WAVE_FENCE();
  const nmx_c2* ht = nmx_w1000_hilbert(ha, hb, T, (const nmx_c2*)A.w1000_tab + NMX_W1000_TW_N, l);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if (!(q < 15 || l < 40)) continue;
    const nmx_c2 h = ht[l + 64 * q];
    const nmx_c2 e = nmx_mk2(nmx_sqrt_fast(y[q].x * y[q].x + h.x * h.x), nmx_sqrt_fast(y[q].y * y[q].y + h.y * h.y));
    __builtin_amdgcn_raw_buffer_store_b64(e, rout, 8 * l + 512 * q, 0, 0);
  }
}
#endif

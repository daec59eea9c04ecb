// nmx_k_specmm.h -- FFT band power (+ Hjorth / LineLength / Raw) of 1000-sample windows with the spectrum on the MATRIX
// pipe and the windows streamed HBM -> LDS by the DMA path: kernel A for BASELINE config[1] and bench.py's Mode A
// (features/oscillatory.py:90-119, hjorth_raw.py:24-42, linelength.py:11-21), the HBM-bound half of the north star.
//
// Arithmetic.  The band features read |X[k]| for at most 32 consecutive bins k (the default bands: 4 .. 34).  With the
// half-sample phase phi_n = 2 pi k (n + 1/2) / 1000 (a unit factor on X[k]: the magnitude does not see it) the kernel of
// the transform is symmetric about n = 499.5 AND, up to (-1)^k, about n = 249.5, so with
//     a = x[n], b = x[499 - n], c = x[500 + n], d = x[999 - n],   n = 0 .. 249,
//     Re Y[k] = sum_n cos(phi_n) ((a + d) + (-1)^k (b + c)),   Im Y[k] = -sum_n sin(phi_n) ((a - d) + (-1)^k (c - b)):
// FOUR real 16 x 250 contractions per window (cos / sin x even / odd k) instead of a 64 x 1000 one, 32 kflop per window,
// on `v_mfma_f32_16x16x4_f32` (exact fp32, its own pipe).  The four runs a, b, c, d of a window are 16-byte-granule
// aligned (that is what the half-sample shift buys: x[n] pairs with x[999 - n], not x[1000 - n]).
//
// Mapping.  One wave = 16 consecutive windows of one channel = the MFMA's 16 columns; the instruction's four k slots are
// four n: lane (j = l & 15, ks = l >> 4) owns n = 32 c + 8 ks + i (i = 0 .. 7) of window j in step c = 0 .. 7.  Per step
// the wave stages, per window, one 128-byte run of each of the four streams (a ascending from 0, b descending from 499,
// c ascending from 500, d descending from 999): 8 KB, eight `global_load_lds_dwordx4` (HBM -> LDS, no registers, each
// instruction eight full 128-byte runs).
//
// Round 5: the TABLE LIVES IN REGISTERS.  A lane only ever multiplies with its own 256 table entries (8 steps x 4 classes
// x 8 n: the MFMA's A operand of lane (row j, slot ks)); at one wave per SIMD the 512-entry register file holds them next
// to the working set, the MFMA reads them where they are, and the LDS belongs to the windows alone: a ring of
// NMX_SMM_RING = 5 step buffers per wave (4 waves x 40 KB = the CU's 160 KB), three to four steps = 24 - 32 KB per wave
// in flight instead of one to two.  With the table went eight of a step's sixteen 16-byte LDS reads; the other eight are
// issued ONE STEP AHEAD into a second register set, so that their latency, too, passes under the arithmetic.  (Measured
// before: the DMA pipeline alone, two steps deep, moved Mode A's 4.2 GB in 0.86 ms; every microsecond of arithmetic
// came on top of that -- profiles/r05_specmm_experiments.txt.)
// The DMA's lane-linear LDS image is swizzled on the SOURCE side (granule q of window j lands at position
// (q + sigma(j)) & 7 of its row) so that every `ds_read_b128` of the consumer is bank-conflict free.  The waves of a
// workgroup share nothing and never synchronise.
// Time domain: the lane that owns 8 consecutive samples of a run also reads a 2-sample halo (left of an ascending run,
// right of a descending one: the neighbour that is already in LDS) and advances its window's sums of u, u^2, d1^2,
// d2^2, |d1| (u = x - pivot, the mean of four samples spread over the window) on PACKED arithmetic (two samples per instruction; the unaligned pairs the differences need
// are one v_pk_mov each) in the shadow of the MFMAs; the four partial sums of a window meet once per tile.
// Conditions (host: nmx_specmm_ok): W = 1000, FFT over the whole window, band means only, bins inside 32 consecutive k
// with k_lo >= 1, no Welch / STFT.  Device only.
#pragma once

#include "nmx_device.h"

// (nmx_common.h: NMX_SMM_NG = 63 table granules of four n -- n < 250 live, 250 / 251 zero --, NMX_SMM_TAB_FLOATS:
// [class: cos even, cos odd, sin even, sin odd][G][row 16][t 4])
#define NMX_SMM_STEP_BYTES 8192                         // 16 windows x 4 streams x 128 bytes
#ifndef NMX_SMM_RING
#define NMX_SMM_RING 5
#endif
#define NMX_SMM_WAVE_BYTES (NMX_SMM_RING * NMX_SMM_STEP_BYTES)
#define NMX_SMM_WAVES 4
#define NMX_SMM_LDS_BYTES (NMX_SMM_WAVES * NMX_SMM_WAVE_BYTES)       // 163 840 = all of it at five slots
static_assert(NMX_SMM_RING >= 3 && NMX_SMM_RING <= 5, "ring depth");

static inline bool nmx_specmm_ok(const NmxTimeOscArgs& A) {
  if (!A.smm_tab || A.W != 1000 || A.n_bands > 8 || A.n_bands < 1) return false;
  if (!A.w500_tab || !A.todo) return false;   // (windows with a NaN / an infinity go to the wave-level kernel)
  if (A.welch.enabled || A.stft.enabled || !A.fft.enabled) return false;
  const NmxOsc& O = A.fft;
  if (O.complex_full || O.estimators != NMXD_EST_MEAN || O.return_spectrum || O.n != 1000) return false;
  if (!(O.k_lo >= 1 && O.k_hi - O.k_lo <= 32 && O.k_hi <= 500 && A.smm_k0 == O.k_lo)) return false;
  // (the 16-byte DMA pieces need 4-byte aligned addresses only -- measured: window starts of every residue mod 4 run at
  // the same rate with the same results -- so the choice of this kernel depends on the plan's SHAPE alone, never on how
  // the hops were batched)
  return true;
}

#ifdef NMX_HOST_EMU
// ---- logic emulator (tests/emu, never loaded by the package): the kernel's ARITHMETIC for one (window, channel) -- the
// twice-folded half-sample-phase contraction against the host's table, band means of log10 |Y| / |Y| over the plan's bin
// ranges, the single-pass time domain about the window's first sample with telescoped difference sums, the overflow
// flag for the wave-level redo -- without its machinery (DMA ring, MFMA operand layout, packed arithmetic): what the
// CPU tier can check is the table, the bin bookkeeping, the formulas and the flag protocol.  fp32 like the device,
// summation orders differ (the tests hold both to the oracle).  -> true: the window is left to the redo pass.
NMX_DEV bool nmx_specmm_item_emu(const NmxTimeOscArgs& A, int w, int c) {
  const NmxOsc& O = A.fft;
  const float* x = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? A.starts[w] : 0ll);
  const float* tab = A.smm_tab;
  const int ke0 = O.k_lo + (O.k_lo & 1), ko0 = O.k_lo + 1 - (O.k_lo & 1);
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, chk = 0.f;
  for (int r = 0; r < 16; ++r)
    for (int par = 0; par < 2; ++par) {
      const int k = (par ? ko0 : ke0) + 2 * r;
      float yc = 0.f, ys = 0.f;
      for (int n = 0; n < 250; ++n) {
        const float a = x[n], b = x[499 - n], cc = x[500 + n], d = x[999 - n];
        const float sc = par ? (a + d) - (b + cc) : (a + d) + (b + cc);
        const float ss = par ? (a - d) - (cc - b) : (a - d) + (cc - b);
        yc += tab[(((size_t)par * NMX_SMM_NG + (size_t)(n >> 2)) * 16 + (size_t)r) * 4 + (size_t)(n & 3)] * sc;
        ys += tab[(((size_t)(2 + par) * NMX_SMM_NG + (size_t)(n >> 2)) * 16 + (size_t)r) * 4 + (size_t)(n & 3)] * ss;
      }
      const float pw = yc * yc + ys * ys;   // (rows past k_hi: zero table rows, pw = 0 -- inside no band)
      chk += pw;
      const float val = O.log_transform ? 0.5f * log10f(pw) : sqrtf(pw);
      for (int b = 0; b < A.n_bands; ++b)
        if (k >= O.bin_lo[b] && k < O.bin_hi[b]) bs[b] += val;
    }
  if (!(chk < INFINITY)) return true;   // a NaN / an infinity among the samples, or a power that overflows
  float* o = A.out + (long long)w * A.n_outputs;
  for (int b = 0; b < A.n_bands; ++b) o[O.cols.base + c * O.cols.ch_stride + b * O.cols.a_stride] = bs[b] * O.inv_bins[b];
  if (A.features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) {
    const float p = 0.25f * ((x[0] + x[999]) + (x[499] + x[500]));   // (the device's pivot)
    // (64 partial sums per statistic, like the device's lanes, then a tree: one fp32 accumulator over 1000 terms would
    // carry sqrt(1000) roundings into the cancellation q0 - s0^2 / W)
    float P[5][64];
    for (int k = 0; k < 5; ++k) for (int l = 0; l < 64; ++l) P[k][l] = 0.f;
    for (int n = 0; n < 1000; ++n) { const float u = x[n] - p; P[0][n & 63] += u; P[1][n & 63] += u * u; }
    for (int n = 0; n < 999; ++n) { const float d1 = (x[n + 1] - p) - (x[n] - p); P[2][n & 63] += d1 * d1; P[4][n & 63] += fabsf(d1); }
    for (int n = 0; n < 998; ++n) {
      const float d2 = ((x[n + 2] - p) - (x[n + 1] - p)) - ((x[n + 1] - p) - (x[n] - p));
      P[3][n & 63] += d2 * d2;
    }
    for (int k = 0; k < 5; ++k)
      for (int h = 32; h >= 1; h >>= 1)
        for (int l = 0; l < h; ++l) P[k][l] += P[k][l + h];
    const float s0 = P[0][0], q0 = P[1][0], q1 = P[2][0], q2 = P[3][0], sa = P[4][0];
    const float u0 = x[0] - p, u1 = x[1] - p, u998 = x[998] - p, u999 = x[999] - p;
    const float rW = 1.f / 1000.f, rW1 = 1.f / 999.f, rW2 = 1.f / 998.f;
    const float sd1 = u999 - u0, sd2 = (u999 - u998) - (u1 - u0);   // telescoped sums of the differences
    const float v0 = (q0 - s0 * s0 * rW) * rW, v1 = (q1 - sd1 * sd1 * rW1) * rW1, v2 = (q2 - sd2 * sd2 * rW2) * rW2;
    const float a0 = v0 < 0.f ? 0.f : v0, a1 = v1 < 0.f ? 0.f : v1, a2 = v2 < 0.f ? 0.f : v2;
    const float act = nmx_clean(a0), mob = nmx_clean(sqrtf(a1 / a0)), comp = nmx_clean(sqrtf(a2 / a1) / mob);
    if (A.features & NMXD_F_HJORTH) {
      const int col = A.hjorth_cols.base + c * A.hjorth_cols.ch_stride;
      o[col] = act;
      o[col + A.hjorth_cols.a_stride] = mob;
      o[col + 2 * A.hjorth_cols.a_stride] = comp;
    }
    if (A.features & NMXD_F_LINELENGTH) o[A.ll_cols.base + c * A.ll_cols.ch_stride] = sa * rW1 * rW1;
    if (A.features & NMXD_F_RAW) o[A.raw_cols.base + c * A.raw_cols.ch_stride] = x[999] + (A.dcf ? A.dcf[c] : 0.f);
  }
  return false;
}
#endif

#ifndef NMX_HOST_EMU
typedef float nmx_v4 __attribute__((ext_vector_type(4)));

// ---- the DMA of one step: eight instructions, LDS destination = M0 + 16 * lane ------------------------------------------
// (inline asm: the compiler's own wait-count pass would put vmcnt(0) in front of every LDS read behind a builtin DMA --
// the whole point is that several steps stay in flight; the waits are counted by hand below.  M0 is compiler-reserved:
// saved and restored.  A compiler-issued vector-memory operation between two of these only makes a counted wait stronger.)
NMX_DEV void nmx_smm_dma8(const char* g0, const char* g1, const char* g2, const char* g3, const char* g4, const char* g5,
                          const char* g6, const char* g7, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"   // (the buffer about to be overwritten: every LDS read of it has returned)
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %9\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_add_u32 m0, %9, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_add_u32 m0, %9, 0x800\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, off\n\t"
      "s_add_u32 m0, %9, 0xc00\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, off\n\t"
      "s_add_u32 m0, %9, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, off\n\t"
      "s_add_u32 m0, %9, 0x1400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %6, off\n\t"
      "s_add_u32 m0, %9, 0x1800\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %7, off\n\t"
      "s_add_u32 m0, %9, 0x1c00\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %8, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5), "v"(g6), "v"(g7), "s"(lds_base)
      : "memory", "scc");
}
// a quarter of it (two instructions: both halves of one stream), for issue points spread over a step
NMX_DEV void nmx_smm_dma2(const char* g0, const char* g1, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_add_u32 m0, %3, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "s"(lds_base)
      : "memory", "scc");
}
template <int N>
NMX_DEV void nmx_smm_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// packed helpers: (a.x +- b.y, a.y +- b.x) -- the descending streams meet the ascending ones pair-reversed
NMX_DEV nmx_c2 nmx_smm_add_sw(nmx_c2 a, nmx_c2 b) {
  nmx_c2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
NMX_DEV nmx_c2 nmx_smm_sub_sw(nmx_c2 a, nmx_c2 b) {
  nmx_c2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a.y, b.x): the pair that straddles two aligned register pairs (one v_pk_mov_b32)
NMX_DEV nmx_c2 nmx_smm_hilo(nmx_c2 a, nmx_c2 b) {
  nmx_c2 r;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a.y - a.x) in both halves
NMX_DEV nmx_c2 nmx_smm_diff(nmx_c2 a) {
  nmx_c2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}

// what a tile leaves for the lanes 0 .. 15 to store (deferred: the stores go out behind the NEXT step's counted wait, so
// that they never sit between a wait and the DMA it counts)
template <int NB>
struct NmxSmmOut {
  float* row;          // output row of the lane's window (NULL: nothing pending)
  unsigned short* flag; // the tile's entry of NmxTimeOscArgs::todo (lane 0; NULL: nothing pending)
  unsigned short dirty;
  int ch;
  float band[NB];
  float act, mob, comp, ll, raw;
};

// per-lane constants of the consumer side (byte offsets inside a step buffer)
struct NmxSmmLane {
  unsigned oa0, oa1;     // ascending run: granules 2 ks, 2 ks + 1
  unsigned od0, od1;     // descending run in ADDRESS order: granules 6 - 2 ks, 7 - 2 ks
  unsigned oha, ohd;     // halos: last two samples of granule 2 ks - 1 / first two of granule 8 - 2 ks (mod 8: ks = 0 reads the previous buffer)
  unsigned ohb0;         // step 0, ks = 0, stream b: x[500], x[501] = stream c, granule 0
  unsigned opilot;       // x[0] of the window: stream a, granule 0 (x[500]: stream c, the same place)
  unsigned opilot7;      // x[499] / x[999]: the last sample of granule 7 of step 0's b / d runs
  int ks;
};

// the samples of one step in registers: the four runs in address order, their halos, (step 0) the window's first sample
struct NmxSmmRegs {
  nmx_v4 a0, a1, b0, b1, c0, c1, d0, d1;
  nmx_c2 ha, hb, hc, hd;
  float pilot;
};

// One tile = 16 consecutive windows of one channel on one wave.
template <int NB, bool TD, bool CLEAN>
struct NmxSmmWave {
  static constexpr int R = NMX_SMM_RING;
  // the plan (kernel-argument segment, ~150 dwords) is re-read with s_load where it is used -- the pointer is laundered
  // there -- instead of being hoisted, with every loop-invariant band mask, into scalar registers that spill
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* KargP;
  KargP Ap;
  unsigned ring;        // this wave's ring (LDS address)
  NmxSmmLane L;
  int lane;
  // DMA side: two source rows per lane (windows (l >> 3) and 8 + (l >> 3) of the tile), granule offset folded in
  const char* src0;
  const char* src1;
  const char* nsrc0;    // the same for the NEXT tile (its first steps are prefetched by this tile's last ones)
  const char* nsrc1;
  unsigned r;           // ring slot of the step being consumed
  NmxSmmOut<NB> pend;
  float tab[8][4][8];   // [step][class][i]: the MFMA A operand of this lane (row j, slot ks), n = 32 step + 8 ks + i

  NMX_DEV const NmxTimeOscArgs& plan() {
    asm volatile("" : "+s"(Ap));
    return *(const NmxTimeOscArgs*)Ap;
  }
  NMX_DEV NmxSmmWave(KargP A_, unsigned lds_, int wave, int lane_) : Ap(A_), lane(lane_) {
    ring = lds_ + (unsigned)wave * NMX_SMM_WAVE_BYTES;
    const int j = lane & 15, ks = lane >> 4;
    const int sg = ((j >> 1) & 1) + 4 * ((j >> 3) & 1);
    auto off = [&](int q) { return (unsigned)((((j >> 3) * 64 + (j & 7) * 8 + ((q + sg) & 7))) * 16); };
    L.ks = ks;
    L.oa0 = off(2 * ks);
    L.oa1 = off(2 * ks + 1);
    L.od0 = off(6 - 2 * ks);
    L.od1 = off(7 - 2 * ks);
    L.oha = off((2 * ks + 7) & 7) + 8;
    L.ohd = off((8 - 2 * ks) & 7);
    L.ohb0 = 2 * 2048 + off(0);
    L.opilot = off(0);
    L.opilot7 = off(7) + 12;
    r = 0;
    pend.row = nullptr;
    pend.flag = nullptr;
    src0 = src1 = nsrc0 = nsrc1 = nullptr;
    // the table: granule G = 8 step + 2 ks + h of class X, row j; n = 252 .. 255 (step 7, ks = 3, h = 1) does not exist:
    // zero factors (the samples under them are real ones -- they belong to stream b -- and finite after cleaning)
    const nmx_v4* tg = (const nmx_v4*)plan().smm_tab;
#pragma unroll
    for (int C = 0; C < 8; ++C)
#pragma unroll
      for (int X = 0; X < 4; ++X)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int G = 8 * C + 2 * ks + h;
          nmx_v4 v = tg[(X * NMX_SMM_NG + (G < NMX_SMM_NG ? G : NMX_SMM_NG - 1)) * 16 + j];
          if (C == 7 && h == 1 && G >= NMX_SMM_NG) v = nmx_v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t2 = 0; t2 < 4; ++t2) tab[C][X][4 * h + t2] = v[t2];
        }
  }

  // source rows of tile t for the DMA lanes (-> nsrc0 / nsrc1)
  NMX_DEV void rows(long long t, int n_windows) {
    const NmxTimeOscArgs& A = plan();
    const int C = A.n_channels;
    const int wb = (int)(t / C), c = (int)(t - (long long)wb * C);
    const int jd = lane >> 3, p = lane & 7, hb = (lane >> 4) & 1;
    int w0 = 16 * wb + jd, w1 = w0 + 8;
    if (w0 >= n_windows) w0 = n_windows - 1;
    if (w1 >= n_windows) w1 = n_windows - 1;
    const float* base = A.x + (long long)c * A.ch_stride;
    const float* r0 = base + (long long)w0 * A.win_stride + (A.starts ? A.starts[w0] : 0ll);
    const float* r1 = base + (long long)w1 * A.win_stride + (A.starts ? A.starts[w1] : 0ll);
    // granule q of the run sits at position p = (q + sigma) & 7: this lane (position p) fetches q = (p - sigma) & 7
    nsrc0 = (const char*)(r0 + 4 * ((p - hb) & 7));
    nsrc1 = (const char*)(r1 + 4 * ((p - hb - 4) & 7));
    // (a table of starts is read with vector loads: the compiler's wait for them also drains the DMAs in flight -- have it
    // HERE, behind a step's counted wait, not wherever the pointers are first used.  Equally spaced windows come without
    // a table: nmx_engine_run.inc)
    asm volatile("" : "+v"(nsrc0), "+v"(nsrc1)::"memory");
  }
  NMX_DEV void adopt() { src0 = nsrc0; src1 = nsrc1; }
  template <int C>
  NMX_DEV void dma(unsigned slot, const char* s0, const char* s1) {
    // byte offsets of the step's runs inside a window: a, b, c, d
    constexpr int oa = 128 * C, ob = 1872 - 128 * C, oc = 2000 + 128 * C, od = 3872 - 128 * C;
    nmx_smm_dma8(s0 + oa, s1 + oa, s0 + ob, s1 + ob, s0 + oc, s1 + oc, s0 + od, s1 + od,
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(ring + slot * NMX_SMM_STEP_BYTES)));
  }

  // stream Q (0 .. 3 = a, b, c, d) of step C alone
  template <int C, int Q>
  NMX_DEV void dma_q(unsigned slot, const char* s0, const char* s1) {
    constexpr int o = Q == 0 ? 128 * C : Q == 1 ? 1872 - 128 * C : Q == 2 ? 2000 + 128 * C : 3872 - 128 * C;
    nmx_smm_dma2(s0 + o, s1 + o, (unsigned)__builtin_amdgcn_readfirstlane((int)(ring + slot * NMX_SMM_STEP_BYTES + 2048u * Q)));
  }
  NMX_DEV static float cl(float v) { return CLEAN ? nmx_clean_bl(v) : v; }
  NMX_DEV nmx_v4 ld4(unsigned addr) const {
    nmx_v4 v = *(__attribute__((address_space(3))) const nmx_v4*)(unsigned long)addr;
    if (CLEAN) { v.x = cl(v.x); v.y = cl(v.y); v.z = cl(v.z); v.w = cl(v.w); }
    return v;
  }
  NMX_DEV nmx_c2 ld2(unsigned addr) const {
    nmx_c2 v = *(__attribute__((address_space(3))) const nmx_c2*)(unsigned long)addr;
    if (CLEAN) { v.x = cl(v.x); v.y = cl(v.y); }
    return v;
  }

  // the samples of step CN: slot `sl` holds them, slot `pv` the step before (halos of the lanes ks = 0)
  template <int CN>
  NMX_DEV void read(unsigned sl, unsigned pv, NmxSmmRegs& N) const {
    const unsigned cur = ring + sl * NMX_SMM_STEP_BYTES, prv = ring + pv * NMX_SMM_STEP_BYTES;
    constexpr bool FIRST = CN == 0;
    const bool k0 = L.ks == 0;
    N.a0 = ld4(cur + L.oa0); N.a1 = ld4(cur + L.oa1);
    N.b0 = ld4(cur + 2048 + L.od0); N.b1 = ld4(cur + 2048 + L.od1);
    N.c0 = ld4(cur + 4096 + L.oa0); N.c1 = ld4(cur + 4096 + L.oa1);
    N.d0 = ld4(cur + 6144 + L.od0); N.d1 = ld4(cur + 6144 + L.od1);
    if (TD) {
      // ks >= 1: the neighbouring granule of this buffer; ks = 0: the previous step's (step 0: a / c have no left
      // neighbour to count -- the seam 499 | 500 belongs to b --, b's right neighbour is c's first granule, d has none)
      const unsigned ba = FIRST ? cur : (k0 ? prv : cur);
      N.ha = ld2(ba + L.oha);
      N.hc = ld2(ba + 4096 + L.oha);
      N.hb = ld2(FIRST && k0 ? cur + L.ohb0 : ba + 2048 + L.ohd);
      N.hd = ld2(ba + 6144 + L.ohd);
    }
    if (FIRST) {
      // the pivot of the single-pass sums: the mean of x[0], x[499], x[500], x[999] -- all in step 0's buffer.  (It was x[0]
      // alone: behind a resampler or a notch the first sample of a window can sit several spreads off the rest -- an edge
      // transient --, and q0 - s0^2 / W then cancels (1 + k^2)-fold: fuzz seed 102790, Hjorth mobility 1.8e-5 off.)
      auto at = [&](unsigned a) { return cl(*(__attribute__((address_space(3))) const float*)(unsigned long)a); };
      N.pilot = 0.25f * ((at(cur + L.opilot) + at(cur + 6144 + L.opilot7)) + (at(cur + 2048 + L.opilot7) + at(cur + 4096 + L.opilot)));
    }
  }

  struct Tile {
    nmx_v4 acc[4];                       // cos even, cos odd, sin even, sin odd: rows 4 ks + reg, column j
    nmx_c2 s0, q0[2], q1[2], q2[2];      // this lane's share of the window's sums, two samples wide
    float sa[2];
    float es0, eq0, eq1, eq2, esa;       // the same from the first and last step (masked, scalar)
    float pilot;
    float u0, u1, u998, u999, xlast;     // lane ks = 0: the window's ends (telescoped sums, Raw)
  };

  // time domain of one 8-sample run e[0 .. 9] = (halo, run) for ascending streams, (run, halo) for descending ones:
  // first / last step of a tile (lane-dependent masks)
  template <bool ASC, bool EDGE /* the window's first / last run: no halo */, bool LASTSTEP>
  NMX_DEV void td_edge(Tile& T, const float (&v)[8], float h0, float h1) const {
    float e[10];
    if (ASC) { e[0] = h0; e[1] = h1; for (int i = 0; i < 8; ++i) e[2 + i] = v[i]; }
    else { for (int i = 0; i < 8; ++i) e[i] = v[i]; e[8] = h0; e[9] = h1; }
    float d1[9], d2[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) d1[k] = e[k + 1] - e[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) d2[k] = d1[k + 1] - d1[k];
    const bool edge = EDGE && L.ks == 0, tail = LASTSTEP && L.ks == 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // ascending: the differences that END at sample i (d1[i + 1], d2[i]); descending: those that START there (d1[i], d2[i])
      bool own = true, m1 = true, m2 = true;
      if (ASC) {
        if (EDGE && edge) { m1 = i >= 1; m2 = i >= 2; }
        if (LASTSTEP && tail) { own = i < 2; m1 = i < 3; m2 = i < 4; }   // n >= 250 belongs to the descending stream; the seam's differences are counted here
      } else {
        if (EDGE && edge) { m1 = i <= 6; m2 = i <= 5; }
        if (LASTSTEP && tail) { own = m1 = m2 = i >= 6; }
      }
      const float u = v[i] - T.pilot;
      const float dd = ASC ? d1[i + 1] : d1[i];
      const float ee = d2[i];
      // (lane-dependent masks: selects, never products -- a masked halo may be any bit pattern)
      T.es0 += own ? u : 0.f;
      T.eq0 += own ? u * u : 0.f;
      T.esa += m1 ? fabsf(dd) : 0.f;
      T.eq1 += m1 ? dd * dd : 0.f;
      T.eq2 += m2 ? ee * ee : 0.f;
    }
  }
  // the same for a run in the interior of the window, two samples per instruction.  X[0..3] = the run in address order,
  // H = its halo; ASC: d1 pairs END at the samples of X[k] (predecessor pairs (x[2k-1], x[2k])), else they START there
  template <bool ASC, int S>
  NMX_DEV void td_mid(Tile& T, const nmx_c2 (&X)[4], nmx_c2 H) const {
    nmx_c2 D[5];
    if (ASC) {
      D[0] = nmx_smm_diff(H);                                   // (., h1 - h0): the difference before the run
      nmx_c2 prev = H;
#pragma unroll
      for (int k = 0; k < 4; ++k) { D[k + 1] = X[k] - nmx_smm_hilo(prev, X[k]); prev = X[k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const nmx_c2 d1 = D[k + 1], d2 = d1 - nmx_smm_hilo(D[k], d1);
        T.q1[S] = nmx_c2_fma(d1, d1, T.q1[S]);
        T.q2[S] = nmx_c2_fma(d2, d2, T.q2[S]);
        T.sa[S] += fabsf(d1.x);
        T.sa[S] += fabsf(d1.y);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) D[k] = nmx_smm_hilo(X[k], k < 3 ? X[k + 1] : H) - X[k];
      D[4] = nmx_smm_diff(H);                                   // (h1 - h0, .): the difference after the run's last one
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const nmx_c2 d1 = D[k], d2 = nmx_smm_hilo(d1, D[k + 1]) - d1;
        T.q1[S] = nmx_c2_fma(d1, d1, T.q1[S]);
        T.q2[S] = nmx_c2_fma(d2, d2, T.q2[S]);
        T.sa[S] += fabsf(d1.x);
        T.sa[S] += fabsf(d1.y);
      }
    }
  }

  // ---- the arithmetic of step C on the registers Rg ---------------------------------------------------------------------
  template <int C>
  NMX_DEV void fold_mfma(Tile& T, const NmxSmmRegs& Rg, nmx_c2 (&U)[4][4]) {
    constexpr bool FIRST = C == 0;
    if (FIRST) T.pilot = Rg.pilot;
#ifdef NMX_SMM_SCALAR   // (experiment: one sample per instruction)
    {
      float va[8], vb[8], vc[8], vd[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = Rg.a0[i]; va[4 + i] = Rg.a1[i]; vb[i] = Rg.b0[i]; vb[4 + i] = Rg.b1[i];
        vc[i] = Rg.c0[i]; vc[4 + i] = Rg.c1[i]; vd[i] = Rg.d0[i]; vd[4 + i] = Rg.d1[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float ua = va[i] - T.pilot, ub = vb[7 - i] - T.pilot, uc = vc[i] - T.pilot, ud = vd[7 - i] - T.pilot;
        const float P = ua + ud, Q = ub + uc, Rr = ua - ud, Uu = uc - ub;
        const float ece = P + Q, eco = P - Q, ese = Rr + Uu, eso = Rr - Uu;
        T.acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][0][i], ece, T.acc[0], 0, 0, 0);
        T.acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][1][i], eco, T.acc[1], 0, 0, 0);
        T.acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][2][i], ese, T.acc[2], 0, 0, 0);
        T.acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][3][i], eso, T.acc[3], 0, 0, 0);
      }
      (void)U;
      return;
    }
#endif
    const nmx_c2 PP = nmx_mk2(T.pilot, T.pilot);
    const nmx_c2 XA[4] = {Rg.a0.xy, Rg.a0.zw, Rg.a1.xy, Rg.a1.zw}, XB[4] = {Rg.b0.xy, Rg.b0.zw, Rg.b1.xy, Rg.b1.zw};
    const nmx_c2 XC[4] = {Rg.c0.xy, Rg.c0.zw, Rg.c1.xy, Rg.c1.zw}, XD[4] = {Rg.d0.xy, Rg.d0.zw, Rg.d1.xy, Rg.d1.zw};
#pragma unroll
    for (int k = 0; k < 4; ++k) { U[0][k] = XA[k] - PP; U[1][k] = XB[k] - PP; U[2][k] = XC[k] - PP; U[3][k] = XD[k] - PP; }
    // fold: n = 32 C + 8 ks + i pairs a[i], b[7 - i] (address order reversed), c[i], d[7 - i]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const nmx_c2 P = nmx_smm_add_sw(U[0][k], U[3][3 - k]), Rr = nmx_smm_sub_sw(U[0][k], U[3][3 - k]);
      const nmx_c2 Q = nmx_smm_add_sw(U[2][k], U[1][3 - k]), Uu = nmx_smm_sub_sw(U[2][k], U[1][3 - k]);
      const nmx_c2 ece = P + Q, eco = P - Q, ese = Rr + Uu, eso = Rr - Uu;
      if (TD && !FIRST && C != 7) T.s0 += ece;   // (the four samples of each n: the window sum)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * k + h;
        T.acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][0][i], h ? ece.y : ece.x, T.acc[0], 0, 0, 0);
        T.acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][1][i], h ? eco.y : eco.x, T.acc[1], 0, 0, 0);
        T.acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][2][i], h ? ese.y : ese.x, T.acc[2], 0, 0, 0);
        T.acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[C][3][i], h ? eso.y : eso.x, T.acc[3], 0, 0, 0);
      }
    }
  }
  template <int C, class Issue>
  NMX_DEV void time_domain(Tile& T, const NmxSmmRegs& Rg, const nmx_c2 (&U)[4][4], Issue issue) {
    if (!TD) { issue(0); issue(1); issue(2); issue(3); return; }
    constexpr bool FIRST = C == 0, LAST = C == 7;
#ifdef NMX_SMM_SCALAR
    constexpr bool MASKED = true;
#else
    constexpr bool MASKED = FIRST || LAST;
#endif
    if (MASKED) {
      float va[8], vb[8], vc[8], vd[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = Rg.a0[i]; va[4 + i] = Rg.a1[i]; vb[i] = Rg.b0[i]; vb[4 + i] = Rg.b1[i];
        vc[i] = Rg.c0[i]; vc[4 + i] = Rg.c1[i]; vd[i] = Rg.d0[i]; vd[4 + i] = Rg.d1[i];
      }
      issue(0);
      td_edge<true, FIRST, LAST>(T, va, Rg.ha.x, Rg.ha.y);
      issue(1);
      td_edge<false, false, LAST>(T, vb, Rg.hb.x, Rg.hb.y);
      issue(2);
      td_edge<true, FIRST, LAST>(T, vc, Rg.hc.x, Rg.hc.y);
      issue(3);
      td_edge<false, FIRST, LAST>(T, vd, Rg.hd.x, Rg.hd.y);
      if (FIRST) {   // lanes ks = 0 hold x[0], x[1] (stream a) and x[998], x[999] (stream d)
        T.u0 = va[0] - T.pilot; T.u1 = va[1] - T.pilot;
        T.u998 = vd[6] - T.pilot; T.u999 = vd[7] - T.pilot; T.xlast = vd[7];
      }
      asm volatile("" : "+v"(T.es0), "+v"(T.eq0), "+v"(T.eq1), "+v"(T.eq2), "+v"(T.esa));
    } else {
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int k = 0; k < 4; ++k) T.q0[st & 1] = nmx_c2_fma(U[st][k], U[st][k], T.q0[st & 1]);
      const nmx_c2 XA[4] = {Rg.a0.xy, Rg.a0.zw, Rg.a1.xy, Rg.a1.zw}, XB[4] = {Rg.b0.xy, Rg.b0.zw, Rg.b1.xy, Rg.b1.zw};
      const nmx_c2 XC[4] = {Rg.c0.xy, Rg.c0.zw, Rg.c1.xy, Rg.c1.zw}, XD[4] = {Rg.d0.xy, Rg.d0.zw, Rg.d1.xy, Rg.d1.zw};
      issue(0);
      td_mid<true, 0>(T, XA, Rg.ha);
      issue(1);
      td_mid<false, 1>(T, XB, Rg.hb);
      issue(2);
      td_mid<true, 0>(T, XC, Rg.hc);
      issue(3);
      td_mid<false, 1>(T, XD, Rg.hd);
      // (the sums are only READ at the end of the tile: left alone, the optimiser sinks every step's time-domain
      // arithmetic down there and parks the samples in scratch until then.  Pin the values here.)
      asm volatile("" : "+v"(T.s0), "+v"(T.q0[0]), "+v"(T.q0[1]), "+v"(T.q1[0]), "+v"(T.q1[1]), "+v"(T.q2[0]), "+v"(T.q2[1]),
                   "+v"(T.sa[0]), "+v"(T.sa[1]));
    }
  }

  // ---- one step of the pipeline -------------------------------------------------------------------------------------------
  //   wait for the DMA of step C + 1 -> issue its LDS reads (-> Nx) -> multiply step C (registers Cu) -> DMA of the step
  //   R ahead into the slot step C leaves -> time domain of step C.  `more`: a tile follows this one (wave-uniform).
  template <int C>
  NMX_DEV void step(Tile& T, NmxSmmRegs& Cu, bool more, long long t_next, int n_windows) {
    constexpr int NAFTER_TAIL = (6 - C) < (R - 2) ? (6 - C < 0 ? 0 : 6 - C) : (R - 2);
    const bool have_next = C < 7 || more;
    // everything older than the newest R - 2 groups has landed: step C + 1 is in its slot
    if (have_next) { if (more) nmx_smm_wait_vm<8 * (R - 2)>(); else nmx_smm_wait_vm<8 * NAFTER_TAIL>(); }
    if (C == 0) flush();
    if (C == 8 - R && more) rows(t_next, n_windows);
    NmxSmmRegs Nx;
    const unsigned rn = r + 1 == (unsigned)R ? 0u : r + 1u;
    if (have_next) read<(C + 1) & 7>(rn, r, Nx);
    __builtin_amdgcn_sched_barrier(0);
    nmx_c2 U[4][4];
#ifndef NMX_SMM_DEBUG_NOCOMP
    fold_mfma<C>(T, Cu, U);
#else
    T.acc[0][0] += Cu.a0.x + Cu.b0.y + Cu.c1.z + Cu.d1.w + tab[C][C & 3][C];   // (experiment: the DMA pipeline alone)
#endif
    // the step R ahead, into the slot this step leaves (its last readers were the halo reads above)
#if defined(NMX_SMM_DMA_SPREAD) && !defined(NMX_SMM_DEBUG_NOCOMP)
    // (two instructions at a time between the blocks of the time domain: an LDS-DMA instruction among VALU work costs a
    // fraction of one in a burst behind MFMAs)
    const bool here = C + R <= 7, there = !here && more;
    const char* q0 = here ? src0 : nsrc0;
    const char* q1 = here ? src1 : nsrc1;
    const unsigned slot_now = r;
    time_domain<C>(T, Cu, U, [&](int k) {
      if (!(here || there)) return;
      if (k == 0) dma_q<(C + R) & 7, 0>(slot_now, q0, q1);
      else if (k == 1) dma_q<(C + R) & 7, 1>(slot_now, q0, q1);
      else if (k == 2) dma_q<(C + R) & 7, 2>(slot_now, q0, q1);
      else dma_q<(C + R) & 7, 3>(slot_now, q0, q1);
    });
#else
    if (C + R <= 7) dma<(C + R) & 7>(r, src0, src1);
    else if (more) dma<(C + R) & 7>(r, nsrc0, nsrc1);
#ifndef NMX_SMM_DEBUG_NOCOMP
    time_domain<C>(T, Cu, U, [](int) {});
#endif
#endif
    r = rn;
    if (have_next) Cu = Nx;
    __builtin_amdgcn_sched_barrier(0);
  }

  NMX_DEV void flush() {
    const NmxTimeOscArgs& A = plan();   // (laundered in uniform control flow: the branches below are per lane)
    if (pend.flag) { *pend.flag = pend.dirty; pend.flag = nullptr; }
    if (!pend.row) return;
    float* o = pend.row;
    const NmxOsc& O = A.fft;
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (b < A.n_bands) o[O.cols.base + pend.ch * O.cols.ch_stride + b * O.cols.a_stride] = pend.band[b];
    if (TD) {
      if (A.features & NMXD_F_HJORTH) {
        const int col = A.hjorth_cols.base + pend.ch * A.hjorth_cols.ch_stride;
        o[col] = pend.act;
        o[col + A.hjorth_cols.a_stride] = pend.mob;
        o[col + 2 * A.hjorth_cols.a_stride] = pend.comp;
      }
      if (A.features & NMXD_F_LINELENGTH) o[A.ll_cols.base + pend.ch * A.ll_cols.ch_stride] = pend.ll;
      if (A.features & NMXD_F_RAW) o[A.raw_cols.base + pend.ch * A.raw_cols.ch_stride] = pend.raw;
    }
    pend.row = nullptr;
  }

  // band means + time-domain features of the finished tile -> pend (lanes 0 .. 15: window j = lane)
  NMX_DEV void finish(Tile& T, long long t, int n_windows) {
    const NmxTimeOscArgs& A = plan();
    const int C = A.n_channels;
    const int wb = (int)(t / C), c = (int)(t - (long long)wb * C);
    const int j = lane & 15, ks = L.ks;
    const int w = 16 * wb + j;
    const NmxOsc& O = A.fft;
    const int ke0 = O.k_lo + (O.k_lo & 1), ko0 = O.k_lo + 1 - (O.k_lo & 1);
    float bs[NB], chk = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) bs[b] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int k = (par ? ko0 : ke0) + 2 * (4 * ks + v);
        const float yc = T.acc[par][v], ys = T.acc[2 + par][v];
        const float pw = yc * yc + ys * ys;
        chk += pw;
        const float val = O.log_transform ? nmx_log10_half_fast(pw) : __builtin_amdgcn_sqrtf(pw);
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (b < A.n_bands) bs[b] += (k >= O.bin_lo[b] && k < O.bin_hi[b]) ? val : 0.f;
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= A.n_bands) continue;
      float tot = bs[b];
      tot += __shfl_xor(tot, 16, 64);
      tot += __shfl_xor(tot, 32, 64);
      pend.band[b] = tot * O.inv_bins[b];
    }
    // A NaN or an infinity among the window's samples reaches every row of its column (0 x NaN = NaN: the empty rows too):
    // the total power says whether the window was clean -- nothing here cleans on load.  Dirty windows are flagged and
    // left to the wave-level kernel with its cleaning path (nmx_kern_timeosc_w1000_todo), as are windows whose power
    // overflows.
    chk += __shfl_xor(chk, 16, 64);
    chk += __shfl_xor(chk, 32, 64);
    // (also behind a stage that has cleaned the samples already: a re-referenced window with members of its group on the
    // rail holds +-1e38s -- its sums of squares are inf - inf here, the wave-level kernel's two-pass forms are not)
    const bool dirty = !CLEAN && !(chk < INFINITY);
    if (TD) {
      float s0 = T.es0 + (T.s0.x + T.s0.y), q0 = T.eq0 + ((T.q0[0].x + T.q0[0].y) + (T.q0[1].x + T.q0[1].y));
      float q1 = T.eq1 + ((T.q1[0].x + T.q1[0].y) + (T.q1[1].x + T.q1[1].y));
      float q2 = T.eq2 + ((T.q2[0].x + T.q2[0].y) + (T.q2[1].x + T.q2[1].y));
      float sa = T.esa + (T.sa[0] + T.sa[1]);
      s0 += __shfl_xor(s0, 16, 64); q0 += __shfl_xor(q0, 16, 64); q1 += __shfl_xor(q1, 16, 64);
      q2 += __shfl_xor(q2, 16, 64); sa += __shfl_xor(sa, 16, 64);
      s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64); q1 += __shfl_xor(q1, 32, 64);
      q2 += __shfl_xor(q2, 32, 64); sa += __shfl_xor(sa, 32, 64);
      const float rW = 1.f / 1000.f, rW1 = 1.f / 999.f, rW2 = 1.f / 998.f;
      const float sd1 = T.u999 - T.u0, sd2 = (T.u999 - T.u998) - (T.u1 - T.u0);   // telescoped sums of the differences
      const float v0 = (q0 - s0 * s0 * rW) * rW;
      const float v1 = (q1 - sd1 * sd1 * rW1) * rW1;
      const float v2 = (q2 - sd2 * sd2 * rW2) * rW2;
      float act, mob, comp;
      const bool normal = v0 > 1e-30f && v0 < 1e30f && v1 > 1e-30f && v1 < 1e30f && v2 < 1e30f && v2 >= 0.f;
      if (normal) {
        act = v0;
        mob = __builtin_amdgcn_sqrtf(v1 * __builtin_amdgcn_rcpf(v0));
        comp = __builtin_amdgcn_sqrtf(v2 * __builtin_amdgcn_rcpf(v1)) * __builtin_amdgcn_rcpf(mob);
      } else {   // flat / degenerate windows: the reference's nan_to_num placement on IEEE arithmetic
        const float a0 = v0 < 0.f ? 0.f : v0, a1 = v1 < 0.f ? 0.f : v1, a2 = v2 < 0.f ? 0.f : v2;
        act = nmx_clean(a0);
        mob = nmx_clean(sqrtf(a1 / a0));
        comp = nmx_clean(sqrtf(a2 / a1) / mob);
      }
      pend.act = act; pend.mob = mob; pend.comp = comp;
      pend.ll = sa * rW1 * rW1;
      pend.raw = T.xlast + (A.dcf ? A.dcf[c] : 0.f);   // (+ the offset the stream was split from)
    }
    pend.ch = c;
    const bool mine = ks == 0 && w < n_windows;
    pend.row = (mine && !dirty) ? A.out + (long long)w * A.n_outputs : nullptr;
    pend.dirty = (unsigned short)(__ballot(mine && dirty) & 0xffffull);   // (lanes 0 .. 15 = windows 0 .. 15 of the tile)
    pend.flag = lane == 0 ? A.todo + t : nullptr;
  }

  NMX_DEV static void clear(Tile& T) {
#pragma unroll
    for (int X = 0; X < 4; ++X) T.acc[X] = nmx_v4{0.f, 0.f, 0.f, 0.f};
    T.s0 = nmx_mk2(0.f, 0.f);
#pragma unroll
    for (int S = 0; S < 2; ++S) { T.q0[S] = T.q1[S] = T.q2[S] = nmx_mk2(0.f, 0.f); T.sa[S] = 0.f; }
    T.es0 = T.eq0 = T.eq1 = T.eq2 = T.esa = 0.f;
    T.pilot = 0.f;
    T.u0 = T.u1 = T.u998 = T.u999 = T.xlast = 0.f;
  }
};
#endif

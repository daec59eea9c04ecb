// nmx_k_specmm.h -- FFT band power (+ Hjorth / LineLength / Raw) of 1000-sample windows with the spectrum on the MATRIX
// pipe and the windows streamed HBM -> LDS by the DMA path: kernel A for BASELINE config[1] and bench.py's Mode A
// (features/oscillatory.py:90-119, hjorth_raw.py:24-42, linelength.py:11-21), the HBM-bound half of the north star.
//
// Arithmetic.  The band features read |X[k]| for at most 32 consecutive bins k (the default bands: 4 .. 34).  With the
// half-sample phase phi_n = 2 pi k (n + 1/2) / 1000 (a unit factor on X[k]: the magnitude does not see it) the kernel of
// the transform is symmetric about n = 499.5 AND, up to (-1)^k, about n = 249.5, so with
//     a = x[n], b = x[499 - n], c = x[500 + n], d = x[999 - n],   n = 0 .. 249,
//     Re Y[k] = sum_n cos(phi_n) ((a + d) + (-1)^k (b + c)),   Im Y[k] = -sum_n sin(phi_n) ((a - d) + (-1)^k (c - b)):
// FOUR real 16 x 250 contractions per window (cos / sin x even / odd k) instead of a 64 x 1000 one -- a quarter of the
// matrix work of the direct sum (round 4's form), 32 kflop per window, on `v_mfma_f32_16x16x4_f32` (exact fp32, its own
// pipe).  The four runs a, b, c, d of a window are 16-byte-granule aligned (that is what the half-sample shift buys:
// x[n] pairs with x[999 - n], not x[1000 - n]).
//
// Mapping.  One wave = 16 consecutive windows of one channel = the MFMA's 16 columns; the instruction's four k slots are
// four n: lane (j = l & 15, ks = l >> 4) owns n = 32 c + 8 ks + i (i = 0 .. 7) of window j in step c = 0 .. 7.  Per step
// the wave stages, per window, one 128-byte run of each of the four streams (a ascending from 0, b descending from 499,
// c ascending from 500, d descending from 999): 8 KB, eight `global_load_lds_dwordx4` (HBM -> LDS, no registers, each
// instruction eight full 128-byte runs), into a ring of three step buffers per wave -- two steps (16 KB per wave, 64 KB
// per CU) in flight under the arithmetic of the current one.  The DMA's lane-linear LDS image is swizzled on the SOURCE
// side (granule q of window j lands at position (q + sigma(j)) & 7 of its row) so that every `ds_read_b128` of the
// consumer is bank-conflict free (model: tools/model_specmm.py).  The 64.5 KB table [class][n / 4][row][n % 4] is staged
// ONCE per persistent workgroup (one per CU, four waves, one per SIMD); after that the waves never synchronise.
// Time domain: the lane that owns 8 consecutive samples of a run also reads a 2-sample halo (left of an ascending run,
// right of a descending one: the neighbour that is already in LDS) and advances its window's sums of u, u^2, d1^2,
// d2^2, |d1| (u = x - x[0]) on the VALU in the shadow of the MFMAs; the four partial sums of a window meet once per tile.
// Conditions (host: nmx_specmm_ok): W = 1000, FFT over the whole window, band means only, bins inside 32 consecutive k
// with k_lo >= 1, no Welch / STFT, window starts multiples of 4 samples.  Device only.
#pragma once

#include "nmx_device.h"

#ifndef NMX_HOST_EMU

// (nmx_common.h: NMX_SMM_NG = 63 table granules of four n -- n < 250 live, 250 / 251 zero --, NMX_SMM_TAB_FLOATS:
// [class: cos even, cos odd, sin even, sin odd][G][row 16][t 4])
#define NMX_SMM_TAB_BYTES (NMX_SMM_TAB_FLOATS * 4)      // 64 512
#define NMX_SMM_STEP_BYTES 8192                         // 16 windows x 4 streams x 128 bytes
#define NMX_SMM_RING 3
#define NMX_SMM_WAVE_BYTES (NMX_SMM_RING * NMX_SMM_STEP_BYTES)
#define NMX_SMM_WAVES 4
#define NMX_SMM_TAB_OFF (NMX_SMM_WAVES * NMX_SMM_WAVE_BYTES)          // the rings first (8 KB aligned), the table behind
#define NMX_SMM_LDS_BYTES (NMX_SMM_TAB_OFF + NMX_SMM_TAB_BYTES)       // 162 816 of 163 840

typedef float nmx_v4 __attribute__((ext_vector_type(4)));
typedef float nmx_v2 __attribute__((ext_vector_type(2)));

static inline bool nmx_specmm_ok(const NmxTimeOscArgs& A) {
  if (!A.smm_tab || A.W != 1000 || A.n_bands > 8 || A.n_bands < 1) return false;
  if (A.welch.enabled || A.stft.enabled || !A.fft.enabled) return false;
  const NmxOsc& O = A.fft;
  if (O.complex_full || O.estimators != NMXD_EST_MEAN || O.return_spectrum || O.n != 1000) return false;
  if (!(O.k_lo >= 1 && O.k_hi - O.k_lo <= 32 && O.k_hi <= 500 && A.smm_k0 == O.k_lo)) return false;
  // the DMA moves 16-byte granules
  return A.starts_mod4 && ((unsigned long long)A.x & 15ull) == 0 && (A.ch_stride & 3) == 0 && (A.win_stride & 3) == 0;
}

// ---- the DMA of one step: eight instructions, LDS destination = M0 + 16 * lane ------------------------------------------
// (inline asm: the compiler's own wait-count pass would put vmcnt(0) in front of every LDS read behind a builtin DMA --
// the whole point is that two steps stay in flight; the waits are counted by hand below.  M0 is compiler-reserved: saved
// and restored.  A compiler-issued vector-memory operation between two of these only makes a counted wait stronger.)
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
#ifndef NMX_SMM_DEBUG_HALFDMA
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
#endif
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5), "v"(g6), "v"(g7), "s"(lds_base)
      : "memory", "scc");
}
template <int N>
NMX_DEV void nmx_smm_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// what a tile leaves for the lanes 0 .. 15 to store (deferred: the stores go out behind the NEXT step's counted wait, so
// that they never sit between a wait and the DMA it counts)
template <int NB>
struct NmxSmmOut {
  float* row;          // output row of the lane's window (NULL: nothing pending)
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
  unsigned opilot;       // x[0] of the window: stream a, granule 0
  unsigned tb;           // table: 512 ks + 16 row
  int ks;
};

// One tile = 16 consecutive windows of one channel on one wave.  `first` / `more`: wave-uniform pipeline state (is this the
// wave's first tile: nothing is in flight yet; is there a tile behind it whose first steps this one prefetches).
template <int NB, bool TD, bool CLEAN>
struct NmxSmmWave {
  // the plan (kernel-argument segment, ~150 dwords) is re-read with s_load where it is used -- the pointer is laundered
  // there -- instead of being hoisted, with every loop-invariant band mask, into scalar registers that spill
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* KargP;
  KargP Ap;
  unsigned lds;         // workgroup LDS base (32-bit LDS address of the dynamic segment)
  unsigned ring;        // this wave's ring (LDS address)
  NmxSmmLane L;
  int lane;
  // DMA side: two source rows per lane (windows (l >> 3) and 8 + (l >> 3) of the tile), granule offset folded in
  const char* src0;
  const char* src1;
  const char* nsrc0;    // the same for the NEXT tile (its first two steps are prefetched by this tile's last two)
  const char* nsrc1;
  unsigned r;           // ring slot of the step being consumed
  NmxSmmOut<NB> pend;

  NMX_DEV const NmxTimeOscArgs& plan() {
    asm volatile("" : "+s"(Ap));
    return *(const NmxTimeOscArgs*)Ap;
  }
  NMX_DEV NmxSmmWave(KargP A_, unsigned lds_, int wave, int lane_) : Ap(A_), lds(lds_), lane(lane_) {
    ring = lds + (unsigned)wave * NMX_SMM_WAVE_BYTES;
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
    L.tb = lds + (unsigned)(512 * ks + 16 * j) + NMX_SMM_TAB_OFF;
    r = 0;
    pend.row = nullptr;
    src0 = src1 = nsrc0 = nsrc1 = nullptr;
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
    // (A.starts is read with vector loads: the compiler's wait for them also drains the DMAs in flight -- have it HERE,
    // behind a step's counted wait, not wherever the pointers are first used)
    asm volatile("" : "+v"(nsrc0), "+v"(nsrc1)::"memory");
  }
  NMX_DEV void adopt() { src0 = nsrc0; src1 = nsrc1; }
  template <int C>
  NMX_DEV void dma(unsigned slot, const char* s0, const char* s1) {
    // byte offsets of the step's runs inside a window: a, b, c, d
#ifdef NMX_SMM_DEBUG_CONTIG   // (experiment: the step's 512 bytes of a window in ONE run -- wrong results)
    constexpr int oa = 512 * C, ob = 512 * C + 128, oc = 512 * C + 256, od = C == 7 ? 3872 : 512 * C + 384;
#else
    constexpr int oa = 128 * C, ob = 1872 - 128 * C, oc = 2000 + 128 * C, od = 3872 - 128 * C;
#endif
    nmx_smm_dma8(s0 + oa, s1 + oa, s0 + ob, s1 + ob, s0 + oc, s1 + oc, s0 + od, s1 + od,
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(ring + slot * NMX_SMM_STEP_BYTES)));
  }

  NMX_DEV static float cl(float v) { return CLEAN ? nmx_clean_bl(v) : v; }
  NMX_DEV static nmx_v4 ldt(unsigned addr) { return *(__attribute__((address_space(3))) const nmx_v4*)(unsigned long)addr; }
  NMX_DEV nmx_v4 ld4(unsigned addr) const {
    nmx_v4 v = ldt(addr);
    if (CLEAN) { v.x = cl(v.x); v.y = cl(v.y); v.z = cl(v.z); v.w = cl(v.w); }
    return v;
  }
  NMX_DEV nmx_v2 ld2(unsigned addr) const {
    nmx_v2 v = *(__attribute__((address_space(3))) const nmx_v2*)(unsigned long)addr;
    if (CLEAN) { v.x = cl(v.x); v.y = cl(v.y); }
    return v;
  }

  struct Tile {
    nmx_v4 acc[4];                       // cos even, cos odd, sin even, sin odd: rows 4 ks + reg, column j
    float s0, q0, q1, q2, sa;            // this lane's share of the window's sums
    float pilot;
    float u0, u1, u998, u999, xlast;     // lane ks = 0: the window's ends (telescoped sums, Raw)
  };

  // time domain of one 8-sample run e[0 .. 9] = (halo, run) for ascending streams, (run, halo) for descending ones
  template <bool ASC, bool EDGE /* the window's first / last run: no halo */, bool LASTSTEP>
  NMX_DEV void td_run(Tile& T, const float (&v)[8], float h0, float h1) const {
    if (!TD) return;
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
      if (EDGE || LASTSTEP) {
        // (lane-dependent masks: selects, never products -- a masked halo may be any bit pattern)
        T.s0 += own ? u : 0.f;
        T.q0 += own ? u * u : 0.f;
        T.sa += m1 ? fabsf(dd) : 0.f;
        T.q1 += m1 ? dd * dd : 0.f;
        T.q2 += m2 ? ee * ee : 0.f;
      } else {
        T.s0 += u;
        T.q0 = fmaf(u, u, T.q0);
        T.sa += fabsf(dd);
        T.q1 = fmaf(dd, dd, T.q1);
        T.q2 = fmaf(ee, ee, T.q2);
      }
    }
  }

  // ---- one step: wait for its data, read it, start the DMA two steps ahead, fold, multiply, time domain -----------------
  // NEXT: the DMA issued here (step C + 2 of the same tile for C < 6, step C - 6 of the next tile otherwise) exists
  template <int C>
  NMX_DEV void step(Tile& T, bool more, long long t_next, int n_windows) {
    const bool dma_ahead = C < 6 || more, next_in_flight = C < 7 || more;
    // the newest eight DMAs (the step after this one) may still be in flight
#ifdef NMX_SMM_DEBUG_HALFDMA   // (experiment: half the bytes per step in flight -- wrong results)
    if (next_in_flight) nmx_smm_wait_vm<4>(); else nmx_smm_wait_vm<0>();
#else
    if (next_in_flight) nmx_smm_wait_vm<8>(); else nmx_smm_wait_vm<0>();
#endif
    if (C == 0) flush();
    if (C == 5 && more) rows(t_next, n_windows);
    const unsigned cur = ring + r * NMX_SMM_STEP_BYTES;
    const unsigned prv = ring + (r == 0 ? 2u : r - 1u) * NMX_SMM_STEP_BYTES;
    constexpr bool FIRST = C == 0, LAST = C == 7;
    const bool k0 = L.ks == 0;
    if (FIRST) T.pilot = cl(*(__attribute__((address_space(3))) const float*)(unsigned long)(cur + L.opilot));
    // the four runs in address order + halos
    float va[8], vb[8], vc[8], vd[8];
    nmx_v2 ha, hb, hc, hd;
    {
      const nmx_v4 a0 = ld4(cur + L.oa0), a1 = ld4(cur + L.oa1);
      const nmx_v4 b0 = ld4(cur + 2048 + L.od0), b1 = ld4(cur + 2048 + L.od1);
      const nmx_v4 c0 = ld4(cur + 4096 + L.oa0), c1 = ld4(cur + 4096 + L.oa1);
      const nmx_v4 d0 = ld4(cur + 6144 + L.od0), d1 = ld4(cur + 6144 + L.od1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = a0[i]; va[4 + i] = a1[i]; vb[i] = b0[i]; vb[4 + i] = b1[i];
        vc[i] = c0[i]; vc[4 + i] = c1[i]; vd[i] = d0[i]; vd[4 + i] = d1[i];
      }
      if (TD) {
        // ks >= 1: the neighbouring granule of this buffer; ks = 0: the previous step's (step 0: a / c have no left
        // neighbour to count -- the seam 499 | 500 belongs to b --, b's right neighbour is c's first granule, d has none)
        const unsigned ba = FIRST ? cur : (k0 ? prv : cur);
        ha = ld2(ba + L.oha);
        hc = ld2(ba + 4096 + L.oha);
        hb = ld2(FIRST && k0 ? cur + L.ohb0 : ba + 2048 + L.ohd);
        hd = ld2(ba + 6144 + L.ohd);
      }
    }
    // table: G = 8 C + 2 ks + h; the last step's ks = 3, h = 1 would be granule 63 (n = 252 .. 255): read 62, data zeroed
    nmx_v4 tab[4][2];
    {
      const unsigned t0 = L.tb + C * 2048;
      const unsigned t1 = (LAST && L.ks == 3) ? t0 : t0 + 256;
#pragma unroll
      for (int X = 0; X < 4; ++X) {
        tab[X][0] = ldt(t0 + X * (NMX_SMM_NG * 256));
        tab[X][1] = ldt(t1 + X * (NMX_SMM_NG * 256));
      }
    }
    // DMA two steps ahead into the slot of the PREVIOUS step (its last reader was the halo read above)
    if (dma_ahead) {
      const unsigned slot = r == 0 ? 2u : r - 1u;
      if (C < 6) dma<(C + 2) & 7>(slot, src0, src1); else dma<(C + 2) & 7>(slot, nsrc0, nsrc1);
    }
#ifdef NMX_SMM_DEBUG_NOCOMP   // (experiment: the streaming rate of the DMA pipeline alone -- wrong results)
    T.acc[0][0] += va[0] + vb[1] + vc[2] + vd[3] + tab[0][0][0];
    r = r == 2 ? 0u : r + 1u;
    __builtin_amdgcn_sched_barrier(0);
    return;
#endif
    // fold: n = 32 C + 8 ks + i pairs a[i], b[7 - i] (address order reversed), c[i], d[7 - i]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float ua = va[i] - T.pilot, ub = vb[7 - i] - T.pilot, uc = vc[i] - T.pilot, ud = vd[7 - i] - T.pilot;
      const float P = ua + ud, Q = ub + uc, R = ua - ud, U = uc - ub;
      float ece = P + Q, eco = P - Q, ese = R + U, eso = R - U;
      if (LAST && i >= 4) {   // n = 252 .. 255 of the lanes ks = 3: the table read was redirected
        const bool dead = L.ks == 3;
        ece = dead ? 0.f : ece; eco = dead ? 0.f : eco; ese = dead ? 0.f : ese; eso = dead ? 0.f : eso;
      }
      T.acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[0][i >> 2][i & 3], ece, T.acc[0], 0, 0, 0);
      T.acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[1][i >> 2][i & 3], eco, T.acc[1], 0, 0, 0);
      T.acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[2][i >> 2][i & 3], ese, T.acc[2], 0, 0, 0);
      T.acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(tab[3][i >> 2][i & 3], eso, T.acc[3], 0, 0, 0);
    }
    if (TD) {
      td_run<true, FIRST, LAST>(T, va, ha.x, ha.y);
      td_run<false, false, LAST>(T, vb, hb.x, hb.y);
      td_run<true, FIRST, LAST>(T, vc, hc.x, hc.y);
      td_run<false, FIRST, LAST>(T, vd, hd.x, hd.y);
      if (FIRST) {   // lanes ks = 0 hold x[0], x[1] (stream a) and x[998], x[999] (stream d)
        T.u0 = va[0] - T.pilot; T.u1 = va[1] - T.pilot;
        T.u998 = vd[6] - T.pilot; T.u999 = vd[7] - T.pilot; T.xlast = vd[7];
      }
    }
    r = r == 2 ? 0u : r + 1u;
    // (the sums are only READ at the end of the tile: left alone, the optimiser sinks every step's time-domain arithmetic
    // down there and parks the samples in scratch until then -- 440 spilled registers.  Pin the values here.)
    if (TD) asm volatile("" : "+v"(T.s0), "+v"(T.q0), "+v"(T.q1), "+v"(T.q2), "+v"(T.sa));
    __builtin_amdgcn_sched_barrier(0);
  }

  NMX_DEV void flush() {
    const NmxTimeOscArgs& A = plan();   // (laundered in uniform control flow: the branch below is per lane)
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
    float bs[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bs[b] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int k = (par ? ko0 : ke0) + 2 * (4 * ks + v);
        const float yc = T.acc[par][v], ys = T.acc[2 + par][v];
        const float pw = yc * yc + ys * ys;
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
    if (TD) {
      float s0 = T.s0, q0 = T.q0, q1 = T.q1, q2 = T.q2, sa = T.sa;
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
      pend.raw = T.xlast;
    }
    pend.ch = c;
    pend.row = (ks == 0 && w < n_windows) ? A.out + (long long)w * A.n_outputs : nullptr;
  }

  NMX_DEV static void clear(Tile& T) {
#pragma unroll
    for (int X = 0; X < 4; ++X) T.acc[X] = nmx_v4{0.f, 0.f, 0.f, 0.f};
    T.s0 = T.q0 = T.q1 = T.q2 = T.sa = 0.f;
    T.pilot = 0.f;
    T.u0 = T.u1 = T.u998 = T.u999 = T.xlast = 0.f;
  }
};
#endif

// nmx_k_sharpwave.h -- kernel E: SharpwaveAnalyzer.analyze_waveform + estimators
// (features/sharpwaves.py:259-328,330-465) for one pre-filtered series per WAVE.
//
// Work per item ((window, channel, filter), both polarities): local extrema with SciPy's
// plateau rule, SciPy's priority-ordered `distance` suppression (greedy by height, solved
// here as a parallel fixed point -- a peak is kept once every higher neighbour within
// `distance` is removed, removed once a higher neighbour is kept), trough <-> peak pairing,
// the 13 per-trough quantities and the estimators.
//
// The kernel is latency bound (a few thousand instructions per series, long dependent chains),
// so the structure minimises dependent LDS round trips and wave-wide reductions:
//   * maxima AND minima are detected in ONE pass over per-lane contiguous chunks held in
//     registers (the "Trough" polarity analyses -y, whose peaks are y's minima: the two raw
//     extrema lists serve all four find_peaks calls of the reference);
//   * the two distance selections of a polarity run in the same fixed-point loop and share its
//     barriers / ballot; peaks without any neighbour inside `distance` are settled up front;
//   * ordered compaction = per-lane chunk + one wave prefix sum (deterministic output order);
//   * index lists are 16-bit, scratch arrays are aliased: ~11 KiB LDS per series.
// Ties of equal height inside `distance`: the later peak wins (SciPy's own order there is
// NumPy's unstable argsort, i.e. implementation-defined).
#pragma once

#include "nmx_device.h"

#include "../../include/nmx.h"  // NMX_SW_* feature ids and NMX_SWE_* estimator ids
#ifdef NMX_HOST_EMU
#include <vector>
#endif

typedef unsigned short nmx_u16;

struct NmxSharpArgs {
  const float* y;   // [n_windows][C][n_filters][W] pre-filtered series
  float* out;
  int n_outputs, n_windows, n_channels, n_filters, W;
  float ms;         // 1000 / sfreq
  int sharp_off;    // int(5 * (1000 / sfreq))
  int dist_peaks, dist_troughs;   // ceil(distance), samples
  int est_peaks, est_troughs;     // polarity enabled flags
  int between;
  int n_combos;
  int combo_feature[NMX_MAX_SW_COMBOS_DEV], combo_est[NMX_MAX_SW_COMBOS_DEV];
  int combo_slot[NMX_MAX_SW_COMBOS_DEV];   // output slot or -1 (num_peaks when between)
  unsigned feature_mask;          // bit f: feature f used by some combo
  NmxCols cols;       // a = filter, b = slot (x n_polarities + polarity when !between)
  NmxCols np_cols;    // num_peaks (between mode), a = filter
  int has_num_peaks;
  int dense_ok;          // distances <= 10 and only mean / max / min estimators: register-resident path
  int fast_estimators;   // all pairs are mean/max/min of loop-free per-trough quantities
  // LDS carve (float offsets): z[W] | emax,emin,selP,selT,lf,rt (u16[pm]) | st (u8[2 pm]) | vals | res | red
  int off_z, off_emax, off_emin, off_selp, off_selt, off_lf, off_rt, off_st, off_vals, off_res, off_red;
  int pm;           // capacity of the index lists (W / 2 + 2)
  int lds_floats;
  // dense-first launch (small LDS footprint, more waves per CU): 128-entry lists, overflowing items
  // are flagged in `todo` and redone by the generic kernel with the full layout above
  int dz_emax, dz_emin, dz_selt, dz_lf, dz_rt, dz_selp, dz_res, dz_lds_floats;
  unsigned char* todo;
};

#ifdef NMX_HOST_EMU
NMX_DEV int nmx_wave_excl_sum_i(int v, int* total) { *total = v; return 0; }
NMX_DEV int nmx_wave_any(int v) { return v; }
#else
// exclusive prefix sum over the wave on the DPP path: Hillis-Steele inside each row of 16
// (row_shr 1, 2, 4, 8), then row_bcast:15 / row_bcast:31 carry the row totals forward
NMX_DEV int nmx_wave_excl_sum_i(int v, int* total) {
  int inc = v;
#define NMX_DPP_I(ctrl, rmask) __builtin_amdgcn_update_dpp(0, inc, ctrl, rmask, 0xf, false)
  inc += NMX_DPP_I(0x111, 0xf);   // row_shr:1
  inc += NMX_DPP_I(0x112, 0xf);   // row_shr:2
  inc += NMX_DPP_I(0x114, 0xf);   // row_shr:4
  inc += NMX_DPP_I(0x118, 0xf);   // row_shr:8
  inc += NMX_DPP_I(0x142, 0xa);   // row_bcast:15
  inc += NMX_DPP_I(0x143, 0xc);   // row_bcast:31
#undef NMX_DPP_I
  *total = __builtin_amdgcn_readlane(inc, 63);
  return inc - v;
}
NMX_DEV int nmx_wave_any(int v) { return __any(v); }
#endif

// ---- extrema detection -----------------------------------------------------------------------
// SciPy _local_maxima_1d on z (maxima) and on -z (minima) in one pass.  Lane chunk [i0, i1);
// a plateau is owned by the lane of its first sample; its midpoint may lie beyond the chunk.
#ifdef NMX_HOST_EMU
NMX_DEV void nmx_extrema(const float* z, int W, nmx_u16* emax, nmx_u16* emin, int* n_max, int* n_min) {
  const int n_idx = W - 2;
  const int chunk = n_idx > 0 ? (n_idx + NMX_NT - 1) / NMX_NT : 0;
  const int i0 = 1 + NMX_TID * chunk;
  const int i1 = (i0 + chunk) < (W - 1) ? (i0 + chunk) : (W - 1);
  int cmax = 0, cmin = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int bmax = 0, bmin = 0;
    if (pass == 1) {
      int total;
      const int packed = cmax | (cmin << 16);
      const int base = nmx_wave_excl_sum_i(packed, &total);
      bmax = base & 0xffff;
      bmin = base >> 16;
      *n_max = total & 0xffff;
      *n_min = total >> 16;
    }
    int kmax = 0, kmin = 0;
    if (i0 < i1) {
      float prev = z[i0 - 1], cur = z[i0];
      for (int i = i0; i < i1; ++i) {
        const float nxt = z[i + 1];
        if (prev < cur) {          // candidate maximum (plateau start)
          int ahead = i + 1;
          float a = nxt;
          while (a == cur && ahead < W - 1) { ++ahead; a = z[ahead]; }
          if (a < cur) {
            if (pass == 1) emax[bmax + kmax] = (nmx_u16)((i + ahead - 1) >> 1);
            ++kmax;
          }
        } else if (prev > cur) {   // candidate minimum
          int ahead = i + 1;
          float a = nxt;
          while (a == cur && ahead < W - 1) { ++ahead; a = z[ahead]; }
          if (a > cur) {
            if (pass == 1) emin[bmin + kmin] = (nmx_u16)((i + ahead - 1) >> 1);
            ++kmin;
          }
        }
        prev = cur;
        cur = nxt;
      }
    }
    if (pass == 0) { cmax = kmax; cmin = kmin; }
  }
  NMX_SYNC();
}
#else
// Device version: the per-lane chunk (<= 64 samples) is classified without branches into two
// (a lane-interleaved variant with ballots instead of the scan -- conflict-free LDS reads -- measured
// 10 % slower: the kernel is issue bound and that form needs more instructions per sample)
// bitmasks (a divergent `if` costs several scalar instructions and the CU has one scalar unit);
// only plateau starts -- rare -- take a branch.  Bit k = extremum whose (plateau) start is i0 + k.
NMX_DEV void nmx_extrema(const float* z, int W, nmx_u16* emax, nmx_u16* emin, int* n_max, int* n_min,
                         int cap = 0x7fffffff) {
  const int n_idx = W - 2;
  const int chunk = n_idx > 0 ? (n_idx + NMX_NT - 1) / NMX_NT : 0;
  const int i0 = 1 + NMX_TID * chunk;
  const int i1 = (i0 + chunk) < (W - 1) ? (i0 + chunk) : (W - 1);
  unsigned long long mmax = 0ull, mmin = 0ull;
  bool no_plateau = false;   // this lane saw no plateau start: every extremum is its own midpoint
  if ((chunk == 16 || chunk == 8) && NMX_NT == 64) {
    // default window (962 < W <= 1026; 450 < W <= 514 at chunk 8: BASELINE config 5): the lane's positions and their two
    // neighbours are chunk + 2 consecutive floats starting at the aligned z[chunk lane] -- 16-byte LDS reads + one 8-byte
    // read instead of chunk + 1 dword reads, and 32-bit masks built from compile-time bit constants
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    float sv[18];
    unsigned m1 = 0u, m2 = 0u;
    bool plateau = false;
    if (chunk == 16) {
      const f4* q4 = (const f4*)(z + 16 * NMX_TID);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4 t = q4[g];
        sv[4 * g] = t.x; sv[4 * g + 1] = t.y; sv[4 * g + 2] = t.z; sv[4 * g + 3] = t.w;
      }
      // (the last lane's tail lies beyond the window: the list / scratch region that follows z in LDS is
      // readable, the values are masked out below)
      const f2 t = *(const f2*)(z + 16 * NMX_TID + 16);
      sv[16] = t.x; sv[17] = t.y;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const bool in = (i0 + k) < i1;
        const float prev = sv[k], cur = sv[k + 1], nxt = sv[k + 2];
        const bool up = in && prev < cur, dn = in && prev > cur;
        m1 |= (up && nxt < cur) ? (1u << k) : 0u;
        m2 |= (dn && nxt > cur) ? (1u << k) : 0u;
        plateau = plateau || ((up || dn) && nxt == cur);
      }
    } else {
      const f4* q4 = (const f4*)(z + 8 * NMX_TID);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f4 t = q4[g];
        sv[4 * g] = t.x; sv[4 * g + 1] = t.y; sv[4 * g + 2] = t.z; sv[4 * g + 3] = t.w;
      }
      const f2 t = *(const f2*)(z + 8 * NMX_TID + 8);
      sv[8] = t.x; sv[9] = t.y;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool in = (i0 + k) < i1;
        const float prev = sv[k], cur = sv[k + 1], nxt = sv[k + 2];
        const bool up = in && prev < cur, dn = in && prev > cur;
        m1 |= (up && nxt < cur) ? (1u << k) : 0u;
        m2 |= (dn && nxt > cur) ? (1u << k) : 0u;
        plateau = plateau || ((up || dn) && nxt == cur);
      }
    }
    if (plateau) {   // rare: resolve plateau starts with the generic walk
      for (int i = i0; i < i1; ++i) {
        const float prev = z[i - 1], cur = z[i], nxt = z[i + 1];
        const bool up = prev < cur, dn = prev > cur;
        if ((up || dn) && nxt == cur) {
          int ahead = i + 1;
          float a = nxt;
          while (a == cur && ahead < W - 1) { ++ahead; a = z[ahead]; }
          if (up && a < cur) m1 |= 1u << (i - i0);
          if (dn && a > cur) m2 |= 1u << (i - i0);
        }
      }
    }
    mmax = m1;
    mmin = m2;
    no_plateau = !plateau;
  } else
  if (chunk > 64) {
    // long windows (W > 4098): the lane's chunk does not fit one 64-bit mask pair -- count, prefix-sum, then walk the
    // chunk again and store (the plateau walk is shared by both passes)
    int cmax = 0, cmin = 0, bmax = 0, bmin = 0;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
        int total;
        const int base = nmx_wave_excl_sum_i(cmax | (cmin << 16), &total);
        bmax = base & 0xffff; bmin = base >> 16;
        *n_max = total & 0xffff;
        *n_min = total >> 16;
      }
      if (i0 < i1) {
        float prev = z[i0 - 1], cur = z[i0];
        for (int i = i0; i < i1; ++i) {
          const float nxt = z[i + 1];
          const bool up = prev < cur, dn = prev > cur;
          bool is_max = up && nxt < cur, is_min = dn && nxt > cur;
          int ahead = i + 1;
          if ((up || dn) && nxt == cur) {   // plateau start (rare)
            float a = nxt;
            while (a == cur && ahead < W - 1) { ++ahead; a = z[ahead]; }
            is_max = up && a < cur;
            is_min = dn && a > cur;
          }
          if (is_max) {
            if (pass == 1) { if (bmax < cap) emax[bmax] = (nmx_u16)((i + ahead - 1) >> 1); ++bmax; } else ++cmax;
          }
          if (is_min) {
            if (pass == 1) { if (bmin < cap) emin[bmin] = (nmx_u16)((i + ahead - 1) >> 1); ++bmin; } else ++cmin;
          }
          prev = cur;
          cur = nxt;
        }
      }
    }
    NMX_SYNC();
    return;
  } else
  if (i0 < i1) {
    float prev = z[i0 - 1], cur = z[i0];
    for (int i = i0; i < i1; ++i) {
      const float nxt = z[i + 1];
      const bool up = prev < cur, dn = prev > cur;
      const unsigned long long bit = 1ull << (i - i0);
      mmax |= (up && nxt < cur) ? bit : 0ull;
      mmin |= (dn && nxt > cur) ? bit : 0ull;
      if ((up || dn) && nxt == cur) {   // plateau start (rare)
        int ahead = i + 1;
        float a = nxt;
        while (a == cur && ahead < W - 1) { ++ahead; a = z[ahead]; }
        if (up && a < cur) mmax |= bit;
        if (dn && a > cur) mmin |= bit;
      }
      prev = cur;
      cur = nxt;
    }
  }
  int total;
  const int base = nmx_wave_excl_sum_i(__popcll(mmax) | (__popcll(mmin) << 16), &total);
  int bmax = base & 0xffff, bmin = base >> 16;
  *n_max = total & 0xffff;
  *n_min = total >> 16;
  while (mmax) {
    const int i = i0 + __ffsll((long long)mmax) - 1;
    mmax &= mmax - 1;
    int ahead = i + 1;
    if (!no_plateau) {
      const float cur = z[i];
      while (ahead < W - 1 && z[ahead] == cur) ++ahead;
    }
    if (bmax < cap) emax[bmax] = (nmx_u16)((i + ahead - 1) >> 1);   // counts stay exact past `cap`
    ++bmax;
  }
  while (mmin) {
    const int i = i0 + __ffsll((long long)mmin) - 1;
    mmin &= mmin - 1;
    int ahead = i + 1;
    if (!no_plateau) {
      const float cur = z[i];
      while (ahead < W - 1 && z[ahead] == cur) ++ahead;
    }
    if (bmin < cap) emin[bmin] = (nmx_u16)((i + ahead - 1) >> 1);
    ++bmin;
  }
  NMX_SYNC();
}
#endif

// ---- distance selection (two problems per call share barriers) -------------------------------
struct NmxSelProb {
  const nmx_u16* pos;   // raw extrema (ascending positions)
  int n;
  float sgn;            // +1: maxima of z, -1: maxima of -z
  int dist;
  unsigned char* st;    // 0 undecided, 1 keep, 2 removed
  nmx_u16* out;         // compacted survivors
  int n_out;
};

NMX_DEV void nmx_select2(const float* z, NmxSelProb* P) {
  // settle peaks without neighbours inside `dist`
  for (int p = 0; p < 2; ++p) {
    const NmxSelProb& Q = P[p];
    for (int j = NMX_TID; j < Q.n; j += NMX_NT) {
      unsigned char s = 1;
      if (Q.dist > 1) {
        const int pj = Q.pos[j];
        const bool nl = (j > 0) && (pj - (int)Q.pos[j - 1] < Q.dist);
        const bool nr = (j + 1 < Q.n) && ((int)Q.pos[j + 1] - pj < Q.dist);
        s = (nl || nr) ? 0 : 1;
      }
      Q.st[j] = s;
    }
  }
  NMX_SYNC();
  for (;;) {
    int undecided = 0;
    for (int p = 0; p < 2; ++p) {
      const NmxSelProb& Q = P[p];
      for (int j = NMX_TID; j < Q.n; j += NMX_NT) {
        if (Q.st[j] != 0) continue;
        const int pj = Q.pos[j];
        const float vj = Q.sgn * z[pj];
        bool removed = false, wait = false;
        for (int k = j - 1; k >= 0 && pj - (int)Q.pos[k] < Q.dist; --k) {
          if (Q.sgn * z[Q.pos[k]] > vj) {  // strictly higher; equal height: later index wins
            const int s = Q.st[k];
            if (s == 1) removed = true; else if (s == 0) wait = true;
          }
        }
        for (int k = j + 1; k < Q.n && (int)Q.pos[k] - pj < Q.dist; ++k) {
          if (Q.sgn * z[Q.pos[k]] >= vj) {
            const int s = Q.st[k];
            if (s == 1) removed = true; else if (s == 0) wait = true;
          }
        }
        if (removed) Q.st[j] = 2;
        else if (!wait) Q.st[j] = 1;
        else undecided = 1;
      }
    }
    NMX_SYNC();
    if (!nmx_wave_any(undecided)) break;
  }
  // ordered compaction of both lists with one prefix sum
  int cnt[2], i0[2], i1[2];
  for (int p = 0; p < 2; ++p) {
    const int chunk = (P[p].n + NMX_NT - 1) / NMX_NT;
    i0[p] = NMX_TID * chunk;
    i1[p] = (i0[p] + chunk) < P[p].n ? (i0[p] + chunk) : P[p].n;
    int c = 0;
    for (int i = i0[p]; i < i1[p]; ++i) c += (P[p].st[i] == 1);
    cnt[p] = c;
  }
  int total;
  const int base = nmx_wave_excl_sum_i(cnt[0] | (cnt[1] << 16), &total);
  const int b[2] = {base & 0xffff, base >> 16};
  P[0].n_out = total & 0xffff;
  P[1].n_out = total >> 16;
  for (int p = 0; p < 2; ++p) {
    int k = 0;
    for (int i = i0[p]; i < i1[p]; ++i)
      if (P[p].st[i] == 1) { P[p].out[b[p] + k] = P[p].pos[i]; ++k; }
  }
  NMX_SYNC();
}

// ---- dense (register-resident) selection + pairing ------------------------------------------------
// When a window has at most 128 maxima and 128 minima (white noise band-limited to 80 Hz at 1 kHz
// gives ~62 of each) every lane OWNS extremum `lane` (slot 0) and `64 + lane` (slot 1) of each kind:
// neighbours come from cross-lane rotations, the keep / undecided sets of the distance suppression
// are pairs of 64-bit ballots, ranks are mbcnt -- no loops over lists, no per-element exec-mask
// branches (the generic list code below costs ~2 300 VALU + 2 000 SALU instructions per item and
// the kernel is instruction-issue bound).  Same-kind extrema are >= 2 samples apart, so with
// distance <= 10 only the 4 nearest neighbours on each side can matter.
#ifndef NMX_HOST_EMU
struct NmxMask128 { unsigned long long a, b; };   // elements 0..63, 64..127

// NSL = 1: windows with at most 64 extrema of each kind -- one slot per lane, K.b is empty (every helper below is
// compiled for both: about half the instructions of the selection / pairing / estimator phases for such a window)
//
// The nine set bits around element e (e - 4 .. e + 4 -> bits 0 .. 8; elements outside the set read 0) come from the
// set shifted up by four elements, five 32-bit words computed on the scalar unit: per lane two selects, one
// v_alignbit and one mask -- the bit-by-bit form took ~50 vector instructions per window, four windows per lane and
// fixed-point step, and made the selection the largest block of the kernel's 2 100 vector instructions per series.
struct NmxPad128 { unsigned p0, p1, p2, p3, p4; };
NMX_DEV NmxPad128 nmx_pad128(const NmxMask128& K) {
  const unsigned a0 = (unsigned)K.a, a1 = (unsigned)(K.a >> 32), b0 = (unsigned)K.b, b1 = (unsigned)(K.b >> 32);
  NmxPad128 P;
  P.p0 = a0 << 4;
  P.p1 = (a1 << 4) | (a0 >> 28);
  P.p2 = (b0 << 4) | (a1 >> 28);
  P.p3 = (b1 << 4) | (b0 >> 28);
  P.p4 = b1 >> 28;
  return P;
}
template <int SL>
NMX_DEV unsigned nmx_win9(const NmxPad128& P, int lane) {   // e = 64 SL + lane
  const bool low = lane < 32;
  const unsigned lo = SL == 0 ? (low ? P.p0 : P.p1) : (low ? P.p2 : P.p3);
  const unsigned hi = SL == 0 ? (low ? P.p1 : P.p2) : (low ? P.p3 : P.p4);
  return __builtin_amdgcn_alignbit(hi, lo, (unsigned)(lane & 31)) & 0x1ffu;
}

// SciPy _select_by_peak_distance as a fixed point: an extremum is removed iff a higher-priority
// neighbour inside the distance is kept, kept iff all of them are removed (hl / hr: those neighbours; bit (4 - d)
// of hl = element e - d, bit (d - 1) of hr = element e + d)
template <int NSL>
NMX_DEV NmxMask128 nmx_dense_fixpoint(const bool* valid, const unsigned* hl, const unsigned* hr, int lane) {
  const unsigned h0 = hl[0] | (hr[0] << 5), h1 = hl[1] | (hr[1] << 5);   // in the layout of nmx_win9
  int s0 = !valid[0] ? 2 : (h0 == 0u ? 1 : 0);
  int s1 = (NSL == 1 || !valid[1]) ? 2 : (h1 == 0u ? 1 : 0);
  for (;;) {
    NmxMask128 K, U;
    K.a = __ballot(s0 == 1); K.b = NSL == 1 ? 0ull : __ballot(s1 == 1);
    U.a = __ballot(s0 == 0); U.b = NSL == 1 ? 0ull : __ballot(s1 == 0);
    if ((U.a | U.b) == 0ull) return K;
    const NmxPad128 PK = nmx_pad128(K), PU = nmx_pad128(U);
    {
      const bool removed = (nmx_win9<0>(PK, lane) & h0) != 0u, wait = (nmx_win9<0>(PU, lane) & h0) != 0u;
      s0 = s0 != 0 ? s0 : (removed ? 2 : (wait ? 0 : 1));
    }
    if (NSL == 2) {
      const bool removed = (nmx_win9<1>(PK, lane) & h1) != 0u, wait = (nmx_win9<1>(PU, lane) & h1) != 0u;
      s1 = s1 != 0 ? s1 : (removed ? 2 : (wait ? 0 : 1));
    }
  }
}

struct NmxDenseSel {
  int pmax[2], pmin[2];       // this lane's maxima / minima positions (slot 0, slot 1)
  // keep masks: [0] maxima @ distance_peaks, [1] minima @ distance_troughs  (polarity 0)
  //             [2] minima @ distance_peaks, [3] maxima @ distance_troughs  (polarity 1)
  NmxMask128 K[4];
};

// neighbours at element distance d of both slots: rotations by d lanes wrap slot 0 into slot 1
template <int NSL, typename T>
NMX_DEV void nmx_dense_nb(const T* v, int d, int lane, T* left, T* right) {
  if (NSL == 1) {   // (wrapped values are never used: the callers test e >= d and e + d < n <= 64)
    left[0] = __shfl(v[0], (lane - d) & 63);
    right[0] = __shfl(v[0], (lane + d) & 63);
    left[1] = left[0]; right[1] = right[0];
    return;
  }
  const T a_dn = __shfl(v[0], (lane - d) & 63), b_dn = __shfl(v[1], (lane - d) & 63);
  const T a_up = __shfl(v[0], (lane + d) & 63), b_up = __shfl(v[1], (lane + d) & 63);
  left[0] = a_dn;                          // element lane - d          (valid when lane >= d)
  left[1] = lane >= d ? b_dn : a_dn;       // element 64 + lane - d
  right[0] = lane + d < 64 ? a_up : b_up;  // element lane + d
  right[1] = b_up;                         // element 64 + lane + d     (valid when lane + d < 64)
}

template <int NSL>
NMX_DEV void nmx_dense_select(const float* z, const nmx_u16* emax, const nmx_u16* emin, int n_max, int n_min,
                              int dp, int dt, NmxDenseSel& D) {
  const int lane = NMX_TID;
  bool vmax[2] = {false, false}, vmin[2] = {false, false};
  float qmax[2] = {0.f, 0.f}, qmin[2] = {0.f, 0.f};   // priorities (heights)
  D.pmax[1] = 0; D.pmin[1] = 0;
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int e = 64 * sl + lane;
    vmax[sl] = e < n_max; vmin[sl] = e < n_min;
    D.pmax[sl] = vmax[sl] ? (int)emax[e] : 0;
    D.pmin[sl] = vmin[sl] ? (int)emin[e] : 0;
    qmax[sl] = vmax[sl] ? z[D.pmax[sl]] : 0.f;
    qmin[sl] = vmin[sl] ? -z[D.pmin[sl]] : 0.f;
  }
  const int md = dp > dt ? dp : dt;
  unsigned hl[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, hr[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
  for (int d = 1; d <= 4; ++d) {
    int aL[2], aR[2], bL[2], bR[2];
    nmx_dense_nb<NSL>(D.pmax, d, lane, aL, aR);
    nmx_dense_nb<NSL>(D.pmin, d, lane, bL, bR);
    bool maxL[2], maxR[2], minL[2], minR[2], any = false;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      const int e = 64 * sl + lane;
      maxL[sl] = vmax[sl] && e >= d; maxR[sl] = vmax[sl] && e + d < n_max;
      minL[sl] = vmin[sl] && e >= d; minR[sl] = vmin[sl] && e + d < n_min;
      any |= (maxL[sl] && D.pmax[sl] - aL[sl] < md) || (maxR[sl] && aR[sl] - D.pmax[sl] < md) ||
             (minL[sl] && D.pmin[sl] - bL[sl] < md) || (minR[sl] && bR[sl] - D.pmin[sl] < md);
    }
    if (!__any(any)) break;   // wave-uniform: farther neighbours are farther away
    float qaL[2], qaR[2], qbL[2], qbR[2];
    nmx_dense_nb<NSL>(qmax, d, lane, qaL, qaR);
    nmx_dense_nb<NSL>(qmin, d, lane, qbL, qbR);
    const unsigned bl = 1u << (4 - d), br = 1u << (d - 1);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      // strictly higher on the left, higher-or-equal on the right: equal heights -> the later one wins
      const bool hAL = maxL[sl] && qaL[sl] > qmax[sl], hAR = maxR[sl] && qaR[sl] >= qmax[sl];
      const bool hBL = minL[sl] && qbL[sl] > qmin[sl], hBR = minR[sl] && qbR[sl] >= qmin[sl];
      const int gaL = D.pmax[sl] - aL[sl], gaR = aR[sl] - D.pmax[sl];
      const int gbL = D.pmin[sl] - bL[sl], gbR = bR[sl] - D.pmin[sl];
      hl[0][sl] |= (hAL && gaL < dp) ? bl : 0u; hr[0][sl] |= (hAR && gaR < dp) ? br : 0u;
      hl[3][sl] |= (hAL && gaL < dt) ? bl : 0u; hr[3][sl] |= (hAR && gaR < dt) ? br : 0u;
      hl[1][sl] |= (hBL && gbL < dt) ? bl : 0u; hr[1][sl] |= (hBR && gbR < dt) ? br : 0u;
      hl[2][sl] |= (hBL && gbL < dp) ? bl : 0u; hr[2][sl] |= (hBR && gbR < dp) ? br : 0u;
    }
  }
  D.K[0] = nmx_dense_fixpoint<NSL>(vmax, hl[0], hr[0], lane);
  D.K[1] = nmx_dense_fixpoint<NSL>(vmin, hl[1], hr[1], lane);
  D.K[2] = nmx_dense_fixpoint<NSL>(vmin, hl[2], hr[2], lane);
  D.K[3] = nmx_dense_fixpoint<NSL>(vmax, hl[3], hr[3], lane);
}

NMX_DEV int nmx_mbcnt64(unsigned long long m) {   // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// compaction of the kept peaks / troughs into selP / selT, pairing (sharpwaves.py:347-374) and the
// (left, right) peak lists; returns the same quantities as the generic code path
template <int NSL>
NMX_DEV void nmx_dense_pair_bisect(const NmxMask128& KP, const int* ppos, const NmxMask128& KT, const int* tpos,
                            nmx_u16* selP, nmx_u16* selT, nmx_u16* lf, nmx_u16* rt,
                            int* nTr_out, int* n_pairs_out, int* first_valid_out, int* nT_out) {
  const int lane = NMX_TID;
  const int nPa = __popcll(KP.a), nTa = __popcll(KT.a);
  const int nPk = nPa + __popcll(KP.b), nTr = nTa + __popcll(KT.b);
  if ((KP.a >> lane) & 1ull) selP[nmx_mbcnt64(KP.a)] = (nmx_u16)ppos[0];
  if (NSL == 2 && ((KP.b >> lane) & 1ull)) selP[nPa + nmx_mbcnt64(KP.b)] = (nmx_u16)ppos[1];
  if ((KT.a >> lane) & 1ull) selT[nmx_mbcnt64(KT.a)] = (nmx_u16)tpos[0];
  if (NSL == 2 && ((KT.b >> lane) & 1ull)) selT[nTa + nmx_mbcnt64(KT.b)] = (nmx_u16)tpos[1];
  NMX_SYNC();
  // number of kept peaks before each of this lane's troughs (nPk <= 64 NSL: 7 or 8 bisection steps)
  int lo[2] = {0, 0};
  unsigned long long L0[2] = {0ull, 0ull}, Vm[2] = {0ull, 0ull};
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int i = 64 * sl + lane;
    const bool has_t = i < nTr;
    const int t = has_t ? (int)selT[i] : 0;
    int l = 0;
#pragma unroll
    for (int step = 64 * NSL; step > 0; step >>= 1) {
      const int idx = l + step;
      l = (idx <= nPk && (int)selP[idx <= nPk ? idx - 1 : 0] < t) ? idx : l;
    }
    lo[sl] = l;
    L0[sl] = __ballot(has_t && l == 0);
    Vm[sl] = __ballot(has_t && l > 0 && l < nPk);
    if (has_t) lf[i] = (nmx_u16)l;   // temporarily the pointer (gathered below)
  }
  const int first_valid = __popcll(L0[0]) + __popcll(L0[1]);
  const int n_pairs = __popcll(Vm[0]) + __popcll(Vm[1]);
  const int lastv = Vm[1] ? 127 - __clzll((long long)Vm[1]) : (Vm[0] ? 63 - __clzll((long long)Vm[0]) : 0);
  const int last_excl = (lastv + 1) < nTr ? (lastv + 1) : nTr;
  NMX_SYNC();
  int lp[2];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int p = 64 * sl + lane;
    lp[sl] = p < n_pairs ? (int)lf[first_valid + p] : 1;
  }
  NMX_SYNC();   // lf[] is overwritten with the left peaks only after every pointer was read
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int p = 64 * sl + lane;
    if (p < n_pairs) {
      rt[p] = selP[lp[sl]];
      lf[p] = selP[lp[sl] - 1];
    }
  }
  NMX_SYNC();
  *nTr_out = nTr; *n_pairs_out = n_pairs; *first_valid_out = first_valid;
  *nT_out = last_excl > first_valid ? last_excl - first_valid : 0;
}
// set bits of K at element indices < r, r in [0, 128]
NMX_DEV int nmx_popc_below128(const NmxMask128& K, int r) {
  const unsigned long long ma = r >= 64 ? ~0ull : ((1ull << r) - 1ull);
  const int rb = r - 64;
  const unsigned long long mb = rb <= 0 ? 0ull : (rb >= 64 ? ~0ull : ((1ull << rb) - 1ull));
  return __popcll(K.a & ma) + __popcll(K.b & mb);
}

// Pairing without searching: maxima and minima of a sequence alternate, so the number of RAW peaks
// before raw trough e is e or e + 1 (whichever kind comes first) and the number of KEPT peaks before
// it is a popcount of the keep mask below that index.  The alternation is verified against the raw
// lists (two independent LDS reads per trough); if it ever fails the bisection version runs instead.
template <int NSL>
NMX_DEV void nmx_dense_pair(const NmxMask128& KP, const int* ppos, const nmx_u16* rawP, int n_rawP,
                            const NmxMask128& KT, const int* tpos, int n_rawT,
                            nmx_u16* selP, nmx_u16* selT, nmx_u16* lf, nmx_u16* rt,
                            int* nTr_out, int* n_pairs_out, int* first_valid_out, int* nT_out) {
  const int lane = NMX_TID;
  const int off = (n_rawP > 0 && n_rawT > 0 && __builtin_amdgcn_readfirstlane(ppos[0]) <
                                                   __builtin_amdgcn_readfirstlane(tpos[0])) ? 1 : 0;
  int lo[2] = {0, 0};
  bool ok = true;
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int e = 64 * sl + lane;
    const int r = e + off;
    const bool has = e < n_rawT;
    if (has) {
      const int t = tpos[sl];
      const bool in = r <= n_rawP;
      const int before = (in && r > 0) ? (int)rawP[r - 1] : -1;
      const int after = (in && r < n_rawP) ? (int)rawP[r] : 0x7fffffff;
      ok = ok && in && before < t && t < after;
    }
    lo[sl] = nmx_popc_below128(KP, r);
  }
  if (!__all(ok)) {   // wave-uniform; never seen on real or synthetic data
    nmx_dense_pair_bisect<NSL>(KP, ppos, KT, tpos, selP, selT, lf, rt, nTr_out, n_pairs_out, first_valid_out, nT_out);
    return;
  }
  const int nPa = __popcll(KP.a), nTa = __popcll(KT.a);
  const int nPk = nPa + __popcll(KP.b), nTr = nTa + __popcll(KT.b);
  if ((KP.a >> lane) & 1ull) selP[nmx_mbcnt64(KP.a)] = (nmx_u16)ppos[0];
  if (NSL == 2 && ((KP.b >> lane) & 1ull)) selP[nPa + nmx_mbcnt64(KP.b)] = (nmx_u16)ppos[1];
  const bool k0 = (KT.a >> lane) & 1ull, k1 = NSL == 2 && ((KT.b >> lane) & 1ull);
  if (k0) { const int rk = nmx_mbcnt64(KT.a); selT[rk] = (nmx_u16)tpos[0]; lf[rk] = (nmx_u16)lo[0]; }
  if (k1) { const int rk = nTa + nmx_mbcnt64(KT.b); selT[rk] = (nmx_u16)tpos[1]; lf[rk] = (nmx_u16)lo[1]; }
  const unsigned long long L0a = __ballot(k0 && lo[0] == 0), L0b = __ballot(k1 && lo[1] == 0);
  NmxMask128 Vm;
  Vm.a = __ballot(k0 && lo[0] > 0 && lo[0] < nPk);
  Vm.b = __ballot(k1 && lo[1] > 0 && lo[1] < nPk);
  const int first_valid = __popcll(L0a) + __popcll(L0b);
  const int n_pairs = __popcll(Vm.a) + __popcll(Vm.b);
  // compacted index of the last valid trough = kept troughs up to and including its raw element - 1
  const int hv = Vm.b ? 127 - __clzll((long long)Vm.b) : (Vm.a ? 63 - __clzll((long long)Vm.a) : -1);
  const int lastv = hv >= 0 ? nmx_popc_below128(KT, hv + 1) - 1 : 0;
  const int last_excl = (lastv + 1) < nTr ? (lastv + 1) : nTr;
  NMX_SYNC();
  int lp[2];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int p = 64 * sl + lane;
    lp[sl] = p < n_pairs ? (int)lf[first_valid + p] : 1;
  }
  NMX_SYNC();   // lf[] is overwritten with the left peaks only after every pointer was read
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int p = 64 * sl + lane;
    if (p < n_pairs) {
      rt[p] = selP[lp[sl]];
      lf[p] = selP[lp[sl] - 1];
    }
  }
  NMX_SYNC();
  *nTr_out = nTr; *n_pairs_out = n_pairs; *first_valid_out = first_valid;
  *nT_out = last_excl > first_valid ? last_excl - first_valid : 0;
}
#endif

// ---- estimators --------------------------------------------------------------------------------
NMX_DEV float nmx_sw_estimate(int est, const float* v, int n, float* red) {
  if (n == 0) return 0.f;  // sharpwaves.py:294: empty -> 0
  switch (est) {
    case NMX_SWE_MEAN: return nmx_est_mean(v, n, red);
    case NMX_SWE_MEDIAN: return nmx_est_median(v, n, red);
    case NMX_SWE_MAX: return nmx_est_max(v, n, red);
    case NMX_SWE_MIN: return nmx_est_min(v, n, red);
    default: {
      const float m = nmx_est_mean(v, n, red);
      const float s = nmx_est_std(v, n, m, red);
      return s * s;
    }
  }
}

NMX_DEV float nmx_sw_pair(int est, float a, float b) {
  switch (est) {
    case NMX_SWE_MEAN:
    case NMX_SWE_MEDIAN: return 0.5f * (a + b);
    case NMX_SWE_MAX: return nmx_nanmax(a, b);
    case NMX_SWE_MIN: return nmx_nanmin(a, b);
    default: { const float m = 0.5f * (a + b); return 0.5f * ((a - m) * (a - m) + (b - m) * (b - m)); }
  }
}

#ifndef NMX_HOST_EMU
// fast estimators of one polarity with at most 64 NSL troughs / pairs: NSL entries per lane
template <int NSL>
NMX_DEV void nmx_sw_fast_est(const NmxSharpArgs& A, const float* z, float sgn, const nmx_u16* trv, const nmx_u16* lf,
                             const nmx_u16* rt, int nT, int n_pairs, int nPT, int W, int s_off, float* res, int pol) {
  // at most two entries per lane: gather the list entries and the samples they point at ONCE,
  // then every (feature, estimator) pair is register arithmetic + one reduction
  int tq[2] = {0, 0}, lq[2] = {0, 0}, rq[2] = {0, 0}, tprev[2] = {0, 0};
  float zt[2], zl[2], zr[2], zm[2], zp[2];
  bool okT[2], okP[2], okS[2];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int p = NMX_TID + 64 * sl;
    okT[sl] = p < nT; okP[sl] = p < n_pairs;
    tq[sl] = okT[sl] ? (int)trv[p] : 0;
    tprev[sl] = (okT[sl] && p > 0) ? (int)trv[p - 1] : tq[sl];
    lq[sl] = okP[sl] ? (int)lf[p] : 0;
    rq[sl] = okP[sl] ? (int)rt[p] : 0;
    okS[sl] = okT[sl] && (tq[sl] - s_off > 0) && (tq[sl] + s_off < W);
    zt[sl] = sgn * z[tq[sl]]; zl[sl] = sgn * z[lq[sl]]; zr[sl] = sgn * z[rq[sl]];
    zm[sl] = okS[sl] ? sgn * z[tq[sl] - s_off] : 0.f;
    zp[sl] = okS[sl] ? sgn * z[tq[sl] + s_off] : 0.f;
  }
  for (int cb = 0; cb < A.n_combos; ++cb) {
    const int f = A.combo_feature[cb], e = A.combo_est[cb];
    if (f == NMX_SW_NUM_PEAKS) continue;
    float acc = e == NMX_SWE_MEAN ? 0.f : (e == NMX_SWE_MAX ? -INFINITY : INFINITY);
    bool has_nan = false;
    int cnt = 0;   // entries that take part: counted with ballots (scalar), not reduced
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      const bool pt = okP[sl] && (NMX_TID + 64 * sl) < nPT;   // arrays that pair troughs with peaks
      float v;
      bool ok;
      switch (f) {
        case NMX_SW_PEAK_LEFT: v = zl[sl]; ok = okP[sl]; break;
        case NMX_SW_PEAK_RIGHT: v = zr[sl]; ok = okP[sl]; break;
        case NMX_SW_TROUGH: v = zt[sl]; ok = okT[sl]; break;
        case NMX_SW_WIDTH: v = (float)(rq[sl] - lq[sl]); ok = okP[sl]; break;
        case NMX_SW_PROMINENCE: v = fabsf((zr[sl] + zl[sl]) * 0.5f - zt[sl]); ok = pt; break;
        case NMX_SW_INTERVAL: v = (float)(tq[sl] - tprev[sl]) * A.ms; ok = okT[sl]; break;
        case NMX_SW_DECAY_TIME: v = (float)(lq[sl] - tq[sl]) * A.ms; ok = pt; break;
        case NMX_SW_RISE_TIME: v = (float)(rq[sl] - tq[sl]) * A.ms; ok = pt; break;
        default: v = zt[sl] - 0.5f * (zm[sl] + zp[sl]); ok = okS[sl]; break;   // NMX_SW_SHARPNESS
      }
      cnt += __popcll(__ballot(ok));
      // max / min through the hardware instruction (it skips a NaN operand); np.max / np.min propagate NaN: one
      // flag per lane, one ballot per estimator -- the NaN-propagating form was five instructions per step of the
      // DPP reduction, 36 per estimator
      has_nan = has_nan || (ok && v != v);
      if (ok) acc = e == NMX_SWE_MEAN ? acc + v : (e == NMX_SWE_MAX ? nmx_vmax(acc, v) : nmx_vmin(acc, v));
    }
    if (e == NMX_SWE_MEAN) acc = nmx_wave_reduce(acc, 0.f, [](float a, float b) { return a + b; });
    else if (e == NMX_SWE_MAX) acc = nmx_wave_reduce(acc, -INFINITY, [](float a, float b) { return nmx_vmax(a, b); });
    else acc = nmx_wave_reduce(acc, INFINITY, [](float a, float b) { return nmx_vmin(a, b); });
    if (e != NMX_SWE_MEAN && __ballot(has_nan)) acc = NAN;
    if (NMX_TID == 0) res[pol * A.n_combos + cb] = cnt == 0 ? 0.f : (e == NMX_SWE_MEAN ? acc / (float)cnt : acc);
  }
}
#endif

// one WAVE per (window, channel, filter)
// LDS working set of one item (pointers, so that the analysis can also run inside the FIR-bank
// kernel on the freshly filtered series: see nmx_k_bank_w64.h)
struct NmxSharpLds {
  float* z;                                       // [W] series
  nmx_u16 *emax, *emin, *selP, *selT, *lf, *rt;   // index lists
  unsigned char* st;                              // selection state (generic path only)
  float *vals, *res, *red;
};

NMX_DEV NmxSharpLds nmx_sharp_layout(const NmxSharpArgs& A, float* smem) {
  NmxSharpLds L;
  L.z = smem + A.off_z;
  L.emax = (nmx_u16*)(smem + A.off_emax);
  L.emin = (nmx_u16*)(smem + A.off_emin);
  L.selP = (nmx_u16*)(smem + A.off_selp);
  L.selT = (nmx_u16*)(smem + A.off_selt);
  L.lf = (nmx_u16*)(smem + A.off_lf);
  L.rt = (nmx_u16*)(smem + A.off_rt);
  L.st = (unsigned char*)(smem + A.off_st);
  L.vals = smem + A.off_vals;
  L.res = smem + A.off_res;   // [2][n_combos] + [2] num_peaks
  L.red = smem + A.off_red;
  return L;
}

// Analysis of the series in L.z.  dense_only: the caller provides lists for 128 entries only; returns
// false (nothing written) when the window needs the generic list code.
#ifdef NMX_SW_PROFILE
#define NMX_SWP(i) { const long long t_ = clock64(); swp[i] += t_ - swl; swl = t_; }
#else
#define NMX_SWP(i)
#endif
NMX_DEV bool nmx_sharp_body(const NmxSharpArgs& A, const NmxSharpLds& L, int w, int c, int fi, bool dense_only,
                            long long t_in = 0) {
#ifdef NMX_SW_PROFILE
  long long swp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, swl = clock64();
  swp[0] = swl - t_in;
#endif
  float* z = L.z;
  nmx_u16 *emax = L.emax, *emin = L.emin, *selP = L.selP, *selT = L.selT, *lf = L.lf, *rt = L.rt;
  unsigned char* st = L.st;
  float *vals = L.vals, *res = L.res, *red = L.red;
  const int W = A.W;
  int n_max = 0, n_min = 0;
#ifdef NMX_HOST_EMU
  nmx_extrema(z, W, emax, emin, &n_max, &n_min);
#else
  // dense_only callers provide lists for 128 entries: longer ones are counted, not stored
  nmx_extrema(z, W, emax, emin, &n_max, &n_min, dense_only ? 128 : 0x7fffffff);
#endif
  NMX_SWP(1)   // extrema
  float* row = A.out + (long long)w * A.n_outputs;
  int pol_slot = 0;
  const int n_pol = (A.est_peaks ? 1 : 0) + (A.est_troughs ? 1 : 0);
#ifndef NMX_HOST_EMU
  // wave-uniform: the register-resident path applies (else the generic list code below)
  const bool dense = A.dense_ok && n_max <= 128 && n_min <= 128;
  if (dense_only && !dense) return false;
  // at most 64 extrema of each kind (wave-uniform): one slot per lane
  const bool one = n_max <= 64 && n_min <= 64;
  NmxDenseSel D;
  if (dense) {
    if (one) nmx_dense_select<1>(z, emax, emin, n_max, n_min, A.dist_peaks, A.dist_troughs, D);
    else nmx_dense_select<2>(z, emax, emin, n_max, n_min, A.dist_peaks, A.dist_troughs, D);
  }
#endif
  NMX_SWP(2)   // distance selection
  for (int pol = 0; pol < 2; ++pol) {
    if ((pol == 0 && !A.est_peaks) || (pol == 1 && !A.est_troughs)) continue;
    const float sgn = pol == 0 ? 1.f : -1.f;   // "Trough" analysis runs on -y
    // peaks of sgn*z with distance_peaks, troughs (= peaks of -sgn*z) with distance_troughs
    int nTr = 0, n_pairs = 0, first_valid = 0, nT = 0;
#ifndef NMX_HOST_EMU
    if (dense && one) {
      nmx_dense_pair<1>(pol == 0 ? D.K[0] : D.K[2], pol == 0 ? D.pmax : D.pmin, pol == 0 ? emax : emin,
                        pol == 0 ? n_max : n_min,
                        pol == 0 ? D.K[1] : D.K[3], pol == 0 ? D.pmin : D.pmax, pol == 0 ? n_min : n_max,
                        selP, selT, lf, rt, &nTr, &n_pairs, &first_valid, &nT);
    } else if (dense) {
      nmx_dense_pair<2>(pol == 0 ? D.K[0] : D.K[2], pol == 0 ? D.pmax : D.pmin, pol == 0 ? emax : emin,
                        pol == 0 ? n_max : n_min,
                        pol == 0 ? D.K[1] : D.K[3], pol == 0 ? D.pmin : D.pmax, pol == 0 ? n_min : n_max,
                        selP, selT, lf, rt, &nTr, &n_pairs, &first_valid, &nT);
    } else
#endif
    {
    NmxSelProb P[2];
    P[0].pos = pol == 0 ? emax : emin; P[0].n = pol == 0 ? n_max : n_min; P[0].sgn = sgn;
    P[0].dist = A.dist_peaks; P[0].st = st; P[0].out = selP; P[0].n_out = 0;
    P[1].pos = pol == 0 ? emin : emax; P[1].n = pol == 0 ? n_min : n_max; P[1].sgn = -sgn;
    P[1].dist = A.dist_troughs; P[1].st = st + A.pm; P[1].out = selT; P[1].n_out = 0;
    nmx_select2(z, P);
    const int nPk = P[0].n_out;
    nTr = P[1].n_out;
    const nmx_u16* pk = selP;
    const nmx_u16* tr = selT;
    // pairing (sharpwaves.py:347-374): ptr = first peak at or after the trough
    int n_leftinv = 0, lastv = 0;
    n_pairs = 0;
    int step0 = 1024;
    while (2 * step0 <= nPk) step0 *= 2;   // wave-uniform; only windows beyond 4092 samples get here
    for (int i = NMX_TID; i < nTr; i += NMX_NT) {
      const int t = tr[i];
      int lo = 0;   // number of peaks before t: branch-free bisection (nPk < 2 * step0)
      for (int step = step0; step > 0; step >>= 1) {
        const int idx = lo + step;
        lo = (idx <= nPk && (int)pk[idx - 1] < t) ? idx : lo;
      }
      lf[i] = (nmx_u16)lo;   // temporarily the pointer
      if (lo == 0) ++n_leftinv;
      else if (lo < nPk) { lastv = i > lastv ? i : lastv; ++n_pairs; }
    }
    // one reduction for the two counters (each < 2^15)
    {
      const int packed = nmx_block_sum_i(n_leftinv | (n_pairs << 16), red);
      n_leftinv = packed & 0xffff;
      n_pairs = packed >> 16;
      lastv = (int)nmx_block_max((float)lastv, red);
    }
    NMX_SYNC();
    first_valid = n_leftinv;
    const int last_excl = (lastv + 1) < nTr ? (lastv + 1) : nTr;
    nT = last_excl > first_valid ? last_excl - first_valid : 0;
    // pointer -> (left, right) peak positions; rt first (lf[] holds the pointers)
    for (int p = NMX_TID; p < n_pairs; p += NMX_NT) rt[p] = pk[lf[first_valid + p]];
    NMX_SYNC();
    {
#ifdef NMX_HOST_EMU
      std::vector<nmx_u16> tmp(n_pairs > 0 ? n_pairs : 1);
      for (int p = 0; p < n_pairs; ++p) tmp[p] = pk[lf[first_valid + p] - 1];
      for (int p = 0; p < n_pairs; ++p) lf[p] = tmp[p];
#else
      // in place: read my entries, barrier, write (lane-strided -> at most pm / 64 registers)
      // (blocks of 1024 pairs: a block reads lf[first_valid + p] at or beyond its own range and writes lf[p] inside it,
      // so an earlier block never overwrites what a later one still has to read)
      for (int base = 0; base < n_pairs; base += 1024) {
        const int end = (base + 1024) < n_pairs ? (base + 1024) : n_pairs;
        nmx_u16 keep[16];
        int kk = 0;
        for (int p = base + NMX_TID; p < end && kk < 16; p += NMX_NT) keep[kk++] = pk[lf[first_valid + p] - 1];
        NMX_SYNC();
        kk = 0;
        for (int p = base + NMX_TID; p < end && kk < 16; p += NMX_NT) lf[p] = keep[kk++];
        if (end < n_pairs) NMX_SYNC();
      }
#endif
    }
    NMX_SYNC();
    }
    NMX_SWP(3)   // pairing
    const nmx_u16* trv = selT + first_valid;  // trough list after the reference's slice
    const int nPT = (n_pairs == nT) ? n_pairs : 0;  // arrays that broadcast pairs with troughs
    if (NMX_TID == 0) res[2 * A.n_combos + pol] = (float)nT;
#ifndef NMX_HOST_EMU
    if (A.fast_estimators) {
      // every (feature, estimator) pair is an associative reduction (mean / max / min) of a
      // per-trough quantity that needs no inner loop: ONE pass over the troughs, no value lists,
      // one shuffle reduction per pair (default settings: 3 pairs)
      const int s_off = A.sharp_off;
      if (nT <= 128 && n_pairs <= 128) {
        if (nT <= 64 && n_pairs <= 64) nmx_sw_fast_est<1>(A, z, sgn, trv, lf, rt, nT, n_pairs, nPT, W, s_off, res, pol);
        else nmx_sw_fast_est<2>(A, z, sgn, trv, lf, rt, nT, n_pairs, nPT, W, s_off, res, pol);
      } else
      for (int cb = 0; cb < A.n_combos; ++cb) {
        const int f = A.combo_feature[cb], e = A.combo_est[cb];
        if (f == NMX_SW_NUM_PEAKS) continue;
        float acc = e == NMX_SWE_MEAN ? 0.f : (e == NMX_SWE_MAX ? -INFINITY : INFINITY);
        int cnt = 0;
        const int n = (f == NMX_SW_PEAK_LEFT || f == NMX_SW_PEAK_RIGHT || f == NMX_SW_WIDTH) ? n_pairs
                      : ((f == NMX_SW_TROUGH || f == NMX_SW_INTERVAL || f == NMX_SW_SHARPNESS) ? nT : nPT);
        for (int p = NMX_TID; p < n; p += NMX_NT) {
          float v;
          bool ok = true;
          switch (f) {
            case NMX_SW_PEAK_LEFT: v = sgn * z[lf[p]]; break;
            case NMX_SW_PEAK_RIGHT: v = sgn * z[rt[p]]; break;
            case NMX_SW_TROUGH: v = sgn * z[trv[p]]; break;
            case NMX_SW_WIDTH: v = (float)((int)rt[p] - (int)lf[p]); break;
            case NMX_SW_PROMINENCE: v = fabsf((sgn * z[rt[p]] + sgn * z[lf[p]]) * 0.5f - sgn * z[trv[p]]); break;
            case NMX_SW_INTERVAL: v = p == 0 ? 0.f : (float)((int)trv[p] - (int)trv[p - 1]) * A.ms; break;
            case NMX_SW_DECAY_TIME: v = (float)((int)lf[p] - (int)trv[p]) * A.ms; break;
            case NMX_SW_RISE_TIME: v = (float)((int)rt[p] - (int)trv[p]) * A.ms; break;
            default: {  // NMX_SW_SHARPNESS
              const int t = trv[p];
              ok = (t - s_off > 0) && (t + s_off < W);
              v = ok ? sgn * z[t] - 0.5f * (sgn * z[t - s_off] + sgn * z[t + s_off]) : 0.f;
            }
          }
          if (ok) {
            ++cnt;
            acc = e == NMX_SWE_MEAN ? acc + v : (e == NMX_SWE_MAX ? nmx_nanmax(acc, v) : nmx_nanmin(acc, v));
          }
        }
        // wave-uniform estimator: one DPP reduction for the value, one for the count
        if (e == NMX_SWE_MEAN) acc = nmx_wave_reduce(acc, 0.f, [](float a, float b) { return a + b; });
        else if (e == NMX_SWE_MAX) acc = nmx_wave_reduce(acc, -INFINITY, [](float a, float b) { return nmx_nanmax(a, b); });
        else acc = nmx_wave_reduce(acc, INFINITY, [](float a, float b) { return nmx_nanmin(a, b); });
        cnt = nmx_wave_reduce(cnt, 0, [](int a, int b) { return a + b; });
        if (NMX_TID == 0) res[pol * A.n_combos + cb] = cnt == 0 ? 0.f : (e == NMX_SWE_MEAN ? acc / (float)cnt : acc);
      }
    } else
#endif
    for (int f = 0; f < NMX_SW_NFEAT; ++f) {
      if (!(A.feature_mask & (1u << f)) || f == NMX_SW_NUM_PEAKS) continue;
      int n = 0;
      NMX_SYNC();
      switch (f) {
        case NMX_SW_PEAK_LEFT: n = n_pairs; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = sgn * z[lf[p]]; break;
        case NMX_SW_PEAK_RIGHT: n = n_pairs; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = sgn * z[rt[p]]; break;
        case NMX_SW_TROUGH: n = nT; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = sgn * z[trv[p]]; break;
        case NMX_SW_WIDTH: n = n_pairs; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = (float)((int)rt[p] - (int)lf[p]); break;
        case NMX_SW_PROMINENCE:
          n = nPT;
          for (int p = NMX_TID; p < n; p += NMX_NT)
            vals[p] = fabsf((sgn * z[rt[p]] + sgn * z[lf[p]]) * 0.5f - sgn * z[trv[p]]);
          break;
        case NMX_SW_INTERVAL:
          n = nT;
          for (int p = NMX_TID; p < n; p += NMX_NT)
            vals[p] = p == 0 ? 0.f : (float)((int)trv[p] - (int)trv[p - 1]) * A.ms;
          break;
        case NMX_SW_DECAY_TIME: n = nPT; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = (float)((int)lf[p] - (int)trv[p]) * A.ms; break;
        case NMX_SW_RISE_TIME: n = nPT; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = (float)((int)rt[p] - (int)trv[p]) * A.ms; break;
        case NMX_SW_SHARPNESS: {
          // ordered compaction of troughs with a +-sharp_off margin (sharpwaves.py:393-406)
          const int s = A.sharp_off;
          const int chunk = (nT + NMX_NT - 1) / NMX_NT;
          const int i0 = NMX_TID * chunk, i1 = (i0 + chunk) < nT ? (i0 + chunk) : nT;
          int cnt = 0;
          for (int i = i0; i < i1; ++i) cnt += ((int)trv[i] - s > 0 && (int)trv[i] + s < W);
          int total;
          const int base = nmx_wave_excl_sum_i(cnt, &total);
          int k = 0;
          for (int i = i0; i < i1; ++i) {
            const int t = trv[i];
            if (t - s > 0 && t + s < W) {
              vals[base + k] = sgn * z[t] - 0.5f * (sgn * z[t - s] + sgn * z[t + s]);
              ++k;
            }
          }
          n = total;
          break;
        }
        case NMX_SW_RISE_STEEPNESS:
        case NMX_SW_DECAY_STEEPNESS:
        case NMX_SW_SLOPE_RATIO:
          n = nPT;
          for (int p = NMX_TID; p < n; p += NMX_NT) {
            float rise = 0.f, decay = 0.f;
            for (int j = lf[p]; j <= (int)trv[p]; ++j) {
              const float d = j > 0 ? fabsf(z[j] - z[j - 1]) : 0.f;
              rise = d > rise ? d : rise;
            }
            for (int j = trv[p]; j <= (int)rt[p]; ++j) {
              const float d = j > 0 ? fabsf(z[j] - z[j - 1]) : 0.f;
              decay = d > decay ? d : decay;
            }
            vals[p] = f == NMX_SW_RISE_STEEPNESS ? rise : (f == NMX_SW_DECAY_STEEPNESS ? decay : rise - decay);
          }
          break;
        default: break;
      }
      NMX_SYNC();
      for (int cb = 0; cb < A.n_combos; ++cb) {
        if (A.combo_feature[cb] != f) continue;
        const float r = nmx_sw_estimate(A.combo_est[cb], vals, n, red);
        if (NMX_TID == 0) res[pol * A.n_combos + cb] = r;
      }
    }
    NMX_SWP(4)   // estimators
    NMX_SYNC();
    if (!A.between && NMX_TID == 0) {
      for (int cb = 0; cb < A.n_combos; ++cb) {
        if (A.combo_slot[cb] < 0) continue;
        const float v = A.combo_feature[cb] == NMX_SW_NUM_PEAKS ? res[2 * A.n_combos + pol]
                                                                : res[pol * A.n_combos + cb];
        row[A.cols.base + c * A.cols.ch_stride + fi * A.cols.a_stride +
            A.combo_slot[cb] * A.cols.b_stride + pol_slot] = v;
      }
    }
    ++pol_slot;
  }
  NMX_SYNC();
  if (A.between && NMX_TID == 0 && n_pol == 2) {
    for (int cb = 0; cb < A.n_combos; ++cb) {
      if (A.combo_slot[cb] < 0) continue;
      row[A.cols.base + c * A.cols.ch_stride + fi * A.cols.a_stride + A.combo_slot[cb] * A.cols.b_stride] =
          nmx_sw_pair(A.combo_est[cb], res[cb], res[A.n_combos + cb]);
    }
    if (A.has_num_peaks)
      row[A.np_cols.base + c * A.np_cols.ch_stride + fi * A.np_cols.a_stride] =
          0.5f * (res[2 * A.n_combos] + res[2 * A.n_combos + 1]);
  }
  NMX_SWP(5)   // outputs
#ifdef NMX_SW_PROFILE
  if (NMX_TID == 0 && fi == 0 && c == 7 && (w == 3 || w == 600))
    printf("[sw w=%d] n_max %d n_min %d | cycles: load %lld extrema %lld select %lld pair %lld est %lld out %lld\n", w, n_max, n_min,
           swp[0], swp[1], swp[2], swp[3], swp[4], swp[5]);
#endif
  return true;
}

NMX_DEV void nmx_sharp_item(const NmxSharpArgs& A, int w, int c, int fi, float* smem) {
  const NmxSharpLds L = nmx_sharp_layout(A, smem);
  const int W = A.W;
  const float* src = A.y + (((long long)w * A.n_channels + c) * A.n_filters + fi) * W;
  float* z = L.z;
  nmx_stage_row(src, W, [=](int i, float v) { z[i] = v; });
  NMX_SYNC();
  nmx_sharp_body(A, L, w, c, fi, false);
}

#ifndef NMX_HOST_EMU
// dense-first variant: compact LDS layout, no list fallback in this launch
NMX_DEV void nmx_sharp_item_dense(const NmxSharpArgs& A, int w, int c, int fi, long long item, float* smem) {
  NmxSharpLds L;
  L.z = smem;
  L.emax = (nmx_u16*)(smem + A.dz_emax); L.emin = (nmx_u16*)(smem + A.dz_emin);
  L.selT = (nmx_u16*)(smem + A.dz_selt); L.lf = (nmx_u16*)(smem + A.dz_lf);
  L.rt = (nmx_u16*)(smem + A.dz_rt); L.selP = (nmx_u16*)(smem + A.dz_selp);
  L.st = nullptr; L.vals = nullptr; L.res = smem + A.dz_res; L.red = nullptr;
  const int W = A.W;
  const float* src = A.y + (((long long)w * A.n_channels + c) * A.n_filters + fi) * W;
  float* z = L.z;
#ifdef NMX_SW_PROFILE
  const long long t_in = clock64();
#else
  const long long t_in = 0;
#endif
  nmx_stage_row(src, W, [=](int i, float v) { z[i] = v; });
  NMX_SYNC();
  const bool done = nmx_sharp_body(A, L, w, c, fi, true, t_in);
  if (NMX_TID == 0) A.todo[item] = done ? 0 : 1;
}
#endif

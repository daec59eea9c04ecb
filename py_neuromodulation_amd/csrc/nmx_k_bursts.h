// nmx_k_bursts.h -- Bursts (features/bursts.py:149-265) after the envelope kernel.
//
// kernel C  nmx_burst_thr_item : one workgroup per (channel, band); walks the windows of the
//           batch in order and maintains the percentile threshold.
// kernel D  nmx_burst_stat_item: one wave per (window, channel, band); threshold -> run
//           statistics -> six outputs.
//
// Threshold semantics (reproduced reference quirk, see oracle/nm_oracle.py
// Bursts.update_threshold for the derivation): bursts.py:171 calls NumPy's private _quantile
// on self.data_buffer, which partitions that buffer IN PLACE; the later "[-n_ring:]" trim
// therefore never discards one of the largest values.  The ring is effectively "the n_ring
// largest-so-far with small ones evicted", and the 'linear' quantile
//     pos = q (m - 1), lo = floor(pos), thr = s[lo] + (pos - lo) (s[lo+1] - s[lo]),
//     m = min(samples appended so far, n_ring)
// only ever reads order statistics counted from the TOP of the whole history:
//     s[lo] = desc[m - 1 - lo],  s[lo+1] = desc[m - 2 - lo].
// So the state per (channel, band) is the descending top-K list of every envelope sample
// appended so far, K = floor((1 - q)(n_ring - 1)) + 2, kept in LDS while a batch runs and in
// HBM between batches (that list + a counter is the only state of the whole engine).
#pragma once

#include "nmx_device.h"
#ifdef NMX_HOST_EMU
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

struct NmxBurstThrArgs {
  const float* env;     // [n_windows][C][Bb][W]
  float* thr;           // [n_windows][C][Bb]
  float* top;           // [C][Bb][K]  descending top-K of the history (state)
  long long* counts;    // [C][Bb] {samples appended so far, windows seen so far} x 2
  int n_windows, n_channels, n_bands, W;
  int K;                // capacity of the top list
  int n_ring;           // int(sfreq * time_duration_s)
  int overlap;          // samples appended per window after the first (bursts.py:81-85)
  double q;             // threshold / 100
  int P2;               // power of two >= max(W, overlap): bitonic sort size of the new piece
  int off_l0, off_l1, off_p, off_red, lds_floats;
  int list_in_global;   // the top-K list stays in its global (L2-resident) state array, LDS holds only the working sets
};

// bitonic sort (descending) of p[0..n2), n2 a power of two, in LDS
NMX_DEV void nmx_bitonic_desc(float* p, int n2) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = NMX_TID; i < n2; i += NMX_NT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float a = p[i], b = p[ixj];
          const bool desc = ((i & k) == 0);
          if (desc ? (a < b) : (a > b)) {
            p[i] = b;
            p[ixj] = a;
          }
        }
      }
      NMX_SYNC();
    }
  }
}

// number of elements of the descending list l[0..n) that are  > v  (strict) / >= v
NMX_DEV int nmx_count_gt(const float* l, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (l[mid] > v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
NMX_DEV int nmx_count_ge(const float* l, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (l[mid] >= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// first index j in ascending ins[0..n) with ins[j] > i  (== number of ins values <= i)
NMX_DEV int nmx_upper_bound_i(const int* ins, int n, int i) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ins[mid] <= i) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// descending rank-by-counting sort of pc[0..n) -> ps (pc padded with -inf to a multiple of 4)
NMX_DEV void nmx_rank_sort_desc(const float* pc, int n, float* ps) {
  const int n4 = (n + 3) & ~3;
  for (int t = NMX_TID; t < n; t += NMX_NT) {
    const float v = pc[t];
    int rank = 0;
#ifndef NMX_HOST_EMU
#pragma unroll 8
#endif
    for (int j = 0; j < n4; j += 4) {
      const float u0 = pc[j], u1 = pc[j + 1], u2 = pc[j + 2], u3 = pc[j + 3];
      rank += (u0 > v) || (u0 == v && j < t);
      rank += (u1 > v) || (u1 == v && j + 1 < t);
      rank += (u2 > v) || (u2 == v && j + 2 < t);
      rank += (u3 > v) || (u3 == v && j + 3 < t);
    }
    ps[rank] = v;
  }
}

// In-place merge of the sorted (descending) ps[0..n_new) into the descending list L[0..len),
// truncated to K entries: every thread stages a contiguous chunk of L in registers and writes
// it back shifted by the number of new samples that sort before it (merge path).  Equal values:
// list entries first.  Barriers inside; returns the new length.
template <int CH>
NMX_DEV int nmx_merge_into(float* L, int len, int K, const float* ps, int* ins, int n_new, float* vals) {
  const int chunk = (K + NMX_NT - 1) / NMX_NT;   // <= CH
  const int i0 = NMX_TID * chunk;
#ifdef NMX_HOST_EMU
  const int CHB = chunk;
#else
  constexpr int CHB = CH;
#endif
  for (int j = NMX_TID; j < n_new; j += NMX_NT) ins[j] = nmx_count_ge(L, len, ps[j]);
  const int i1 = (i0 + chunk) < len ? (i0 + chunk) : len;
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
  for (int k = 0; k < CHB; ++k) {
    if (k < chunk && i0 + k < i1) vals[k] = L[i0 + k];
  }
  NMX_SYNC();
  if (i0 < i1) {
    int cnt = nmx_upper_bound_i(ins, n_new, i0);
    const int cnt_end = nmx_upper_bound_i(ins, n_new, i1 - 1);
    if (cnt == cnt_end) {   // common case: no new sample lands inside my chunk
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int k = 0; k < CHB; ++k) {
        const int i = i0 + k;
        if (k < chunk && i < i1 && i + cnt < K) L[i + cnt] = vals[k];
      }
    } else {
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int k = 0; k < CHB; ++k) {
        const int i = i0 + k;
        if (k < chunk && i < i1) {
          while (cnt < n_new && ins[cnt] <= i) ++cnt;
          const int pos = i + cnt;
          if (pos < K) L[pos] = vals[k];
        }
      }
    }
  }
  for (int j = NMX_TID; j < n_new; j += NMX_NT) {
    const int pos = ins[j] + j;
    if (pos < K) L[pos] = ps[j];
  }
  NMX_SYNC();
  return (len + n_new) < K ? (len + n_new) : K;
}

NMX_DEV float nmx_lerp_thr(double a, double b, double frac, bool have_b) {
  if (!have_b) return (float)a;
  const double d = b - a;
  return (float)((frac >= 0.5) ? b - d * (1.0 - frac) : a + d * frac);  // NumPy _lerp
}

#define NMX_THR_F 512     // capacity of the sorted low fringe
#define NMX_THR_FREFILL 384
#define NMX_THR_P 1024    // capacity of the pending (unsorted) list

// Threshold walk of one (channel, band) over the hops of a batch.
//
// Fill regime (fewer than n_ring samples so far): every hop merges its sorted new samples into
// the descending top-K list L (in-place merge path, ~4 barriers).
//
// Steady regime (ring full; the long-run state): the rank that is read is pinned a fixed
// distance above the SMALLEST kept value, a new sample only matters if it exceeds that
// smallest value, and each accepted sample evicts the current smallest.  The kept set is held
// as  L_main (sorted, untouched)  +  P (pending accepted samples above the fringe, unsorted)
//   +  F (the few hundred smallest kept values, sorted ascending).
// A hop classifies its <= `overlap` new samples against F's ends, appends to P or inserts into
// F, drops the a smallest of F and reads the threshold from F -- a handful of tiny steps; only
// when F runs low or P fills up (every ~10 hops) P is sorted and merged into L_main and F is
// re-cut from the list's tail.  Bit-identical to the per-hop full merge.
template <int CH>
NMX_DEV void nmx_burst_thr_item(const NmxBurstThrArgs& A, int c, int bi, float* smem) {
  float* L = A.list_in_global ? A.top + ((long long)c * A.n_bands + bi) * A.K : smem + A.off_l0;
  float* pc = smem + A.off_p;            // [P2] raw new samples / flush staging
  float* ps = pc + A.P2;                 // [P2] sorted new samples
  int* ins = (int*)(ps + A.P2);          // [P2] insertion indices
  float* F = (float*)(ins + A.P2);       // [NMX_THR_F] ascending fringe
  float* F2 = F + NMX_THR_F;
  float* Pp = F2 + NMX_THR_F;            // [NMX_THR_P] pending
  int* cnts = (int*)(Pp + NMX_THR_P);    // [4] LDS counters
  const int K = A.K, W = A.W;
  const long long sidx = (long long)c * A.n_bands + bi;
  long long total = A.counts[2 * sidx];
  long long nwin = A.counts[2 * sidx + 1];
  int len = (int)(total < K ? total : K);
  float* gtop = A.top + sidx * K;
  if (!A.list_in_global) for (int i = NMX_TID; i < len; i += NMX_NT) L[i] = gtop[i];
#ifdef NMX_HOST_EMU
  std::vector<float> vals_store(K > 0 ? K : 1);
  float* vals = vals_store.data();
#else
  float vals[CH];
#endif
  // steady-state bookkeeping
  const bool steady_ok = A.P2 >= NMX_THR_P && K > 2 * NMX_THR_F && A.overlap <= 256;
  bool steady = false;
  int Lm = 0, nF = 0, nP = 0;
  const long long m_ring = A.n_ring;
  const double pos_ring = A.q * (double)(m_ring - 1);
  const long long lo_ring = (long long)floor(pos_ring);
  const double frac_ring = pos_ring - (double)lo_ring;
  const int ia_ring = (int)(m_ring - 1 - lo_ring);   // descending index of s[lo]

  const bool can_prefetch = A.overlap <= NMX_NT;
  float pre = 0.f;
  bool have_pre = false;
  NMX_SYNC();
  for (int w = 0; w < A.n_windows; ++w) {
    const int n_new = (nwin == 0) ? W : A.overlap;
    const int n4 = (n_new + 3) & ~3;
    const float* e = A.env + (((long long)w * A.n_channels + c) * A.n_bands + bi) * W + (W - n_new);
    if (have_pre) {
      if (NMX_TID < n_new) pc[NMX_TID] = pre;
    } else {
      for (int i = NMX_TID; i < n_new; i += NMX_NT) pc[i] = e[i];
    }
    for (int i = n_new + NMX_TID; i < n4; i += NMX_NT) pc[i] = -INFINITY;
    have_pre = false;
    if (can_prefetch && w + 1 < A.n_windows) {
      const float* en = A.env + (((long long)(w + 1) * A.n_channels + c) * A.n_bands + bi) * W + (W - A.overlap);
      if (NMX_TID < A.overlap) pre = en[NMX_TID];
      have_pre = true;
    }
    const bool want_steady = steady_ok && total >= m_ring && len == K && ia_ring >= K - 3 && ia_ring < K;
    if (want_steady && !steady) {   // enter: cut the fringe from the list's tail
      nF = NMX_THR_FREFILL;
      Lm = K - nF;
      nP = 0;
      NMX_SYNC();
      for (int j = NMX_TID; j < nF; j += NMX_NT) F[j] = L[K - 1 - j];
      steady = true;
#ifdef NMX_HOST_EMU
      if (getenv("NMX_EMU_TRACE")) fprintf(stderr, "[emu] steady regime entered at hop %d (c=%d b=%d)\n", w, c, bi);
#endif
    }
    if (NMX_TID == 0) { cnts[0] = 0; cnts[1] = 0; cnts[2] = 0; }
    NMX_SYNC();
    if (!steady) {
      nmx_rank_sort_desc(pc, n_new, ps);
      NMX_SYNC();
      len = nmx_merge_into<CH>(L, len, K, ps, ins, n_new, vals);
      total += n_new;
      nwin += 1;
      if (NMX_TID == 0) {
        const long long m = total < m_ring ? total : m_ring;
        const double pos = A.q * (double)(m - 1);
        const long long lo = (long long)floor(pos);
        A.thr[((long long)w * A.n_channels + c) * A.n_bands + bi] =
            nmx_lerp_thr((double)L[m - 1 - lo], lo + 1 <= m - 1 ? (double)L[m - 2 - lo] : 0.0,
                         pos - (double)lo, lo + 1 <= m - 1);
      }
      continue;
    }
    // ---------------- steady hop ----------------
    const float T = F[0], Fmax = F[nF - 1];
    for (int t = NMX_TID; t < n_new; t += NMX_NT) {
      const float x = pc[t];
      if (x > T) {
#ifdef NMX_HOST_EMU
        if (x <= Fmax) ps[cnts[1]++] = x; else Pp[nP + cnts[2]++] = x;
        cnts[0]++;
#else
        if (x <= Fmax) ps[atomicAdd(&cnts[1], 1)] = x; else Pp[nP + atomicAdd(&cnts[2], 1)] = x;
        atomicAdd(&cnts[0], 1);
#endif
      }
    }
    NMX_SYNC();
    const int a = cnts[0], nI = cnts[1];
    nP += cnts[2];
    // new fringe = (F u I) minus its a smallest; ties: fringe entries first
    for (int i = NMX_TID; i < nF; i += NMX_NT) {
      const float v = F[i];
      int lt = 0;
      for (int r = 0; r < nI; ++r) lt += (ps[r] < v);
      const int idx = i + lt - a;
      if (idx >= 0) F2[idx] = v;
    }
    for (int r = NMX_TID; r < nI; r += NMX_NT) {
      const float v = ps[r];
      int rank = 0;   // among the insert candidates (ascending, stable)
      for (int j = 0; j < nI; ++j) rank += (ps[j] < v) || (ps[j] == v && j < r);
      int le = 0;     // fringe entries <= v  (binary search, F ascending)
      { int lo2 = 0, hi2 = nF; while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (F[mid] <= v) lo2 = mid + 1; else hi2 = mid; } le = lo2; }
      const int idx = rank + le - a;
      if (idx >= 0) F2[idx] = v;
    }
    nF = nF + nI - a;
    { float* tsw = F; F = F2; F2 = tsw; }
    total += n_new;
    nwin += 1;
    NMX_SYNC();
    if (NMX_TID == 0) {
      // s[lo] = kept value with descending index ia_ring -> ascending fringe index K - 1 - ia_ring
      const int ja = K - 1 - ia_ring;
      A.thr[((long long)w * A.n_channels + c) * A.n_bands + bi] =
          nmx_lerp_thr((double)F[ja], (double)F[ja + 1], frac_ring, true);
    }
    // flush when the fringe could run dry or the pending list could overflow on the next hop
    if (nF < A.overlap + 8 || nP + A.overlap > NMX_THR_P || w + 1 == A.n_windows) {
      NMX_SYNC();
      // sort pending (descending) through pc -> ps, then merge into L_main
      for (int i = NMX_TID; i < nP; i += NMX_NT) pc[i] = Pp[i];
      for (int i = nP + NMX_TID; i < ((nP + 3) & ~3); i += NMX_NT) pc[i] = -INFINITY;
      NMX_SYNC();
      nmx_rank_sort_desc(pc, nP, ps);
      NMX_SYNC();
      Lm = nmx_merge_into<CH>(L, Lm, K, ps, ins, nP, vals);
      for (int j = NMX_TID; j < nF; j += NMX_NT) L[Lm + j] = F[nF - 1 - j];
      NMX_SYNC();
      // (Lm + nF == K by construction)
      nF = NMX_THR_FREFILL;
      Lm = K - nF;
      nP = 0;
      for (int j = NMX_TID; j < nF; j += NMX_NT) F[j] = L[K - 1 - j];
      NMX_SYNC();
    }
  }
  NMX_SYNC();
  if (!A.list_in_global) for (int i = NMX_TID; i < len; i += NMX_NT) gtop[i] = L[i];
  if (NMX_TID == 0) {
    A.counts[2 * sidx] = total;
    A.counts[2 * sidx + 1] = nwin;
  }
}

#ifndef NMX_HOST_EMU
// ---------------------------------------------------------------------------------------
// Steady regime of the threshold walk with ONE WAVE per (channel, band).
//
// The 256-thread kernel above spends ~1.5 us per hop in workgroup barriers, LDS atomics and one-hop-deep
// global loads, and its 8 waves x 169 VGPRs + 100 KB of LDS per CU starve the throughput kernels that
// run next to it.  When the ring is already full at the first hop of a batch (the host knows: it counts the
// hops) the same algorithm runs here without any barrier:
//   * the new samples of a hop are two registers per lane (overlap <= 128), loaded NMX_THRW_PF hops ahead;
//   * classification is two ballots; a hop without a sample above the smallest kept value (most hops once
//     the history is long) costs a ballot, a compare and the store of the unchanged threshold;
//   * fringe update and the pending list use mbcnt compaction and wave-local LDS fences;
//   * the sorted top-K list never leaves its global state array (L2): a flush streams it once, block by
//     block from the top, shifting every entry up by the number of pending samples that sort before it
//     (binary search in the sorted pending list in LDS) -- all loads independent, no dependent chains.
// 17 KB of LDS and one wave per sequence.  Same values as nmx_burst_thr_item, bit for bit.
#define NMX_THRW_PF 16
#define NMX_THRW_I 256
// + K / 64 + 2 block counters (launcher)
// NR = registers per lane holding a hop's new samples: 2 (overlap <= 128) or 4 (overlap <= 256: 2 kHz at 10 Hz)
#define NMX_THRW_PF_LL 8   // with the list in LDS: 8 hops ahead (4 KB less: THREE walks per CU, 1536 series in two rounds)
#define NMX_THRW_PF_OF(NR, LL) ((NR) == 2 ? ((LL) ? NMX_THRW_PF_LL : NMX_THRW_PF) : 4)
#define NMX_THRW_LDS_FLOATS_OF(NR, LL) (2 * 256 * (NR) + 3 * ((NR) == 2 ? NMX_THR_P : 512) + NMX_THRW_I + NMX_THRW_PF_OF(NR, LL) * 64 * (NR))
#define NMX_THRW_LDS_FLOATS_NR(NR) NMX_THRW_LDS_FLOATS_OF(NR, false)
#define NMX_THRW_LDS_FLOATS NMX_THRW_LDS_FLOATS_NR(2)

// number of entries of the DESCENDING list l[0..n) that are >= v
NMX_DEV int nmx_count_ge_lds(const float* l, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (l[mid] >= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// number of entries of the DESCENDING list l[0..n) that are > v
NMX_DEV int nmx_count_gt_lds(const float* l, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (l[mid] > v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// host-side test: may the wave kernel take this batch?  (first hop already in the steady regime)
static inline bool nmx_burst_thr_wave_ok(const NmxBurstThrArgs& A, long long windows_seen) {
  const int nr = A.overlap <= 128 ? 2 : 4;   // registers per lane -> fringe capacity 256 nr, refill 192 nr
  if (windows_seen <= 0 || A.overlap > 256 || A.overlap + 8 >= 192 * nr) return false;
  const long long total = (long long)A.W + (windows_seen - 1) * (long long)A.overlap;
  const long long m_ring = A.n_ring;
  const double pos_ring = A.q * (double)(m_ring - 1);
  const long long lo_ring = (long long)floor(pos_ring);
  const int ia_ring = (int)(m_ring - 1 - lo_ring);
  return A.K > 2 * 256 * nr && total >= m_ring && ia_ring >= A.K - 3 && ia_ring < A.K;
}

#ifdef NMX_THRW_PROFILE
#define NMX_TP(i) { const long long t_ = clock64(); tp[i] += t_ - tlast; tlast = t_; }
#else
#define NMX_TP(i)
#endif
// Bitonic sort of 64 R values held R per lane (element e = lane + 64 r), DESCENDING in e.  Every loop is compile time:
// a partner at distance >= 64 is another register of the same lane (a max / min pair, no predicate), a nearer one is the
// same register of lane ^ j (one cross-lane read); the direction of a step is a lane bit below 64 and a compile-time
// register bit above.  256 values: 36 steps, 33 of them with a cross-lane read -- ~4 k cycles where ranking by counting
// (every value against every other) took 45 k per flush.
#ifndef NMX_THRW_BITONIC_MAX
#define NMX_THRW_BITONIC_MAX 512   // (0: rank by counting always -- A/B builds)
#endif
template <int R>
NMX_DEV void nmx_bitonic_desc(float (&v)[R], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int jr = j >> 6;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r & jr) continue;
          const bool up = ((64 * r) & k) == 0;   // (k >= 128 here: a register bit)
          const float a = v[r], b = v[r | jr];
          const float hi = fmaxf(a, b), lo = fminf(a, b);
          v[r] = up ? hi : lo;
          v[r | jr] = up ? lo : hi;
        }
      } else {
        const bool lower = (lane & j) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool up = k < 64 ? (lane & k) == 0 : ((64 * r) & k) == 0;
          const float a = v[r], b = __shfl_xor(a, j, 64);
          v[r] = (lower == up) ? fmaxf(a, b) : fminf(a, b);
        }
      }
    }
  }
}

// LL: the top-K list itself lives in LDS for the duration of the launch (copied in at entry, out at exit) whenever it fits
// next to the working set twice per CU (K <= ~13 500: the default 30 s history at 1 kHz is 7 500 entries, 30 KB).  A flush
// then merges at LDS speed: through the L2-resident array it was 104 k cycles of dependent global round trips per flush
// (profiles/r04_stream_trace.txt: 53 % of a young stream's walk, which flushes every ~15 hops).
template <int NR, bool LL>
NMX_DEV void nmx_burst_thr_wave_item(const NmxBurstThrArgs& A, int c, int bi, float* smem) {
  const int lane = (int)(threadIdx.x & 63);
  // fringe capacity scales with the samples a hop brings: a stationary signal accepts ~(1 - q) of them, each
  // evicts one fringe entry, and a flush (sort + stream of the whole top-K list) is due when the fringe runs low
  constexpr int TF = 256 * NR, TREFILL = 192 * NR;
  // NR = 4 (2 kHz hops, thousands of series): a shorter pending list and a shorter prefetch group bring the wave to
  // 20 KB of LDS -- eight walks per CU, one round for 2048 series instead of two
  constexpr int TP = NR == 2 ? NMX_THR_P : 512, PF = NMX_THRW_PF_OF(NR, LL);
#ifdef NMX_THRW_PROFILE
  long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
  int n_ins = 0, n_flush = 0;
#endif
  float* F = smem;                          // [TF] ascending fringe: entries F[fh .. fh + nF)
  float* F2 = F + TF;
  float* Pp = F2 + TF;               // [TP] pending, unsorted
  float* ps = Pp + TP;               // [TP] pending, sorted descending (flush)
  int* ins = (int*)(ps + TP);        // [TP] insertion indices (flush)
  float* I = (float*)(ins + TP);     // [NMX_THRW_I] this hop's fringe inserts
  float* stage = I + NMX_THRW_I;            // [PF][64 NR] new samples of the current group of hops
  int* cb = (int*)(stage + PF * 64 * NR);   // [K / 64 + 2] pending samples above each 64-entry block (flush)
  const int K = A.K, W = A.W, ov = A.overlap;
  const long long sidx = (long long)c * A.n_bands + bi;
  float* Lg = A.top + sidx * K;             // descending top-K list (global, L2 resident)
  float* L = LL ? (float*)(cb + (K / 64 + 4)) : Lg;
  if (LL) {
    for (int j0 = 0; j0 < K; j0 += 64 * 8) {   // eight independent loads in flight
      float t8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int j = j0 + 64 * q + lane; t8[q] = j < K ? Lg[j] : 0.f; }
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int j = j0 + 64 * q + lane; if (j < K) L[j] = t8[q]; }
    }
    NMX_WAVE_FENCE();
  }
  long long total = A.counts[2 * sidx];
  long long nwin = A.counts[2 * sidx + 1];
  const long long m_ring = A.n_ring;
  const double pos_ring = A.q * (double)(m_ring - 1);
  const long long lo_ring = (long long)floor(pos_ring);
  const double frac_ring = pos_ring - (double)lo_ring;
  const int ja = K - 1 - (int)(m_ring - 1 - lo_ring);   // ascending fringe index of s[lo]: 0, 1 or 2

  // enter: cut the fringe from the list's tail
  int nF = TREFILL, fh = 0, Lm = K - nF, nP = 0;
  for (int j = lane; j < nF; j += 64) F[j] = L[K - 1 - j];
  NMX_WAVE_FENCE();
  float T = F[0], Fmax = F[nF - 1];
  float thr_cur = nmx_lerp_thr((double)F[ja], (double)F[ja + 1], frac_ring, true);

  const long long hop_stride = (long long)A.n_channels * A.n_bands * W;
  const float* e0 = A.env + ((long long)c * A.n_bands + bi) * W + (W - ov);
  float nx[NR][PF];   // samples of the next PF hops, register q of a lane = sample lane + 64 q
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const float* e = e0 + (long long)u * hop_stride;
    const bool in = u < A.n_windows;
#pragma unroll
    for (int q = 0; q < NR; ++q) nx[q][u] = (in && lane + 64 * q < ov) ? e[lane + 64 * q] : -INFINITY;
  }
  for (int w0 = 0; w0 < A.n_windows; w0 += PF) {
    // the group's samples go to LDS so that the hop loop below stays ROLLED (one copy of the insert and
    // flush code)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int q = 0; q < NR; ++q) stage[64 * NR * u + 64 * q + lane] = nx[q][u];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // in flight while this group of hops is processed
      const int w = w0 + PF + u;
      const float* e = e0 + (long long)w * hop_stride;
      const bool in = w < A.n_windows;
#pragma unroll
      for (int q = 0; q < NR; ++q) nx[q][u] = (in && lane + 64 * q < ov) ? e[lane + 64 * q] : -INFINITY;
    }
    NMX_WAVE_FENCE();
    NMX_TP(0)   // group top: wait for the prefetched samples, stage them, issue the next loads
    const int ng = (A.n_windows - w0) < PF ? (A.n_windows - w0) : PF;
#pragma nounroll
    for (int u = 0; u < ng; ++u) {
      const int w = w0 + u;
      float x[NR];
      bool cc[NR];
      unsigned long long bm[NR];
      int a = 0;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        x[q] = stage[64 * NR * u + 64 * q + lane];
        cc[q] = x[q] > T;
        bm[q] = __ballot(cc[q]);
        a += __popcll(bm[q]);
      }
      NMX_TP(1)   // classify
      if (a) {
        // accepted samples either fall inside the fringe (inserts, list I) or above it (pending list P);
        // both lists are filled in sample order: register q before q + 1, lanes ascending (mbcnt compaction)
        const unsigned long long below = (1ull << lane) - 1ull;
        bool ii[NR];
        unsigned long long bim[NR];
        int nI = 0, np_new = 0;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          ii[q] = cc[q] && x[q] <= Fmax;
          bim[q] = __ballot(ii[q]);
          const unsigned long long bpm = bm[q] & ~bim[q];
          if (cc[q] && !ii[q]) Pp[nP + np_new + __popcll(bpm & below)] = x[q];
          np_new += __popcll(bpm);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) nI += __popcll(bim[q]);
        nP += np_new;
        if (nI == 0) {
          // every accepted sample lies above the fringe: the a smallest fringe entries are evicted and
          // nothing moves -- advance the head
          fh += a;
          nF -= a;
        } else {
          {
            int at = 0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
              if (ii[q]) I[at + __popcll(bim[q] & below)] = x[q];
              at += __popcll(bim[q]);
            }
          }
          NMX_WAVE_FENCE();
          // new fringe = (F u I) minus its a smallest; ties: fringe entries first
          const float* Fc = F + fh;
          if (nI <= 8) {
            // the inserts fit in eight wave-uniform registers -> no dependent LDS chains.  Fringe entry i
            // moves to i + #(inserts < F[i]) - a; insert r lands at #(inserts before it) + #(F <= I[r]) - a,
            // the second count taken with ballots in the same pass.
            float iv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) iv[r] = r < nI ? I[r] : INFINITY;
            float fv[TF / 64];
#pragma unroll
            for (int q = 0; q < TF / 64; ++q) fv[q] = (lane + 64 * q < nF) ? Fc[lane + 64 * q] : INFINITY;
            int le[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) le[r] = 0;
#pragma unroll
            for (int q = 0; q < TF / 64; ++q) {
              if (64 * q >= nF) break;
              const float v = fv[q];
              int lt = 0;
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                lt += (iv[r] < v);
                le[r] += __popcll(__ballot(v <= iv[r]));   // (padding lanes hold +inf)
              }
              const int idx = lane + 64 * q + lt - a;
              if (lane + 64 * q < nF && idx >= 0) F2[idx] = v;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              if (r >= nI) break;
              int rank = 0;   // among the insert candidates (ascending, stable)
#pragma unroll
              for (int j = 0; j < 8; ++j) rank += (iv[j] < iv[r]) || (iv[j] == iv[r] && j < r);
              const int idx = rank + le[r] - a;
              if (lane == 0 && idx >= 0) F2[idx] = iv[r];
            }
          } else {
            for (int i = lane; i < nF; i += 64) {
              const float v = Fc[i];
              int lt = 0;
              for (int r = 0; r < nI; ++r) lt += (I[r] < v);
              const int idx = i + lt - a;
              if (idx >= 0) F2[idx] = v;
            }
            for (int r = lane; r < nI; r += 64) {
              const float v = I[r];
              int rank = 0;   // among the insert candidates (ascending, stable)
              for (int j = 0; j < nI; ++j) rank += (I[j] < v) || (I[j] == v && j < r);
              int lo2 = 0, hi2 = nF;   // fringe entries <= v
              while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (Fc[mid] <= v) lo2 = mid + 1; else hi2 = mid; }
              const int idx = rank + lo2 - a;
              if (idx >= 0) F2[idx] = v;
            }
          }
          nF = nF + nI - a;
          fh = 0;
          { float* tsw = F; F = F2; F2 = tsw; }
          NMX_WAVE_FENCE();
        }
        T = F[fh];
        thr_cur = nmx_lerp_thr((double)F[fh + ja], (double)F[fh + ja + 1], frac_ring, true);
#ifdef NMX_THRW_PROFILE
        ++n_ins;
#endif
        NMX_TP(2)   // fringe update
      }
      total += ov;
      nwin += 1;
      if (lane == 0) A.thr[((long long)w * A.n_channels + c) * A.n_bands + bi] = thr_cur;
      // flush when the fringe could run dry or the pending list could overflow on the next hop
      if (nF < ov + 8 || nP + ov > TP || w + 1 == A.n_windows) {
        NMX_TP(3)   // threshold store + bookkeeping
        // (1) pending -> ps, sorted descending (rank by counting; equal values: lower index first)
        const int n4 = (nP + 3) & ~3;
        for (int i = nP + lane; i < n4; i += 64) Pp[i] = -INFINITY;
        NMX_WAVE_FENCE();
        // (equal samples are the same float: their order does not show in the sorted list)
        if (LL && nP > 128 && nP <= NMX_THRW_BITONIC_MAX) {   // (the network costs the same for 10 samples as for 512: a steady
          float sv[8];                                         //  stream's short lists, and its kernel, keep the counting form)
#pragma unroll
          for (int r = 0; r < 8; ++r) sv[r] = (lane + 64 * r < nP) ? Pp[lane + 64 * r] : -INFINITY;
          nmx_bitonic_desc<8>(sv, lane);
#pragma unroll
          for (int r = 0; r < 8; ++r) if (lane + 64 * r < nP) ps[lane + 64 * r] = sv[r];
        } else
        for (int t0 = 0; t0 < nP; t0 += 64) {
          // rank of (v, t) in the order "larger value first, equal values: lower index first".  For the
          // lanes of one round the index test is wave-uniform outside the round's own 64 entries:
          // j < t0 -> count u >= v,  j >= t0 + 64 -> count u > v  (two instructions per entry)
          const int t = t0 + lane;
          const float v = t < nP ? Pp[t] : 0.f;
          int rank = 0;
#pragma unroll 2
          for (int j = 0; j < t0; j += 4) {
            const float u0 = Pp[j], u1 = Pp[j + 1], u2 = Pp[j + 2], u3 = Pp[j + 3];
            rank += (u0 >= v); rank += (u1 >= v); rank += (u2 >= v); rank += (u3 >= v);
          }
          const int t1 = (t0 + 64) < n4 ? (t0 + 64) : n4;
          for (int j = t0; j < t1; j += 4) {
            const float u0 = Pp[j], u1 = Pp[j + 1], u2 = Pp[j + 2], u3 = Pp[j + 3];
            rank += (u0 > v) || (u0 == v && j < t);
            rank += (u1 > v) || (u1 == v && j + 1 < t);
            rank += (u2 > v) || (u2 == v && j + 2 < t);
            rank += (u3 > v) || (u3 == v && j + 3 < t);
          }
#pragma unroll 2
          for (int j = t1; j < n4; j += 4) {
            const float u0 = Pp[j], u1 = Pp[j + 1], u2 = Pp[j + 2], u3 = Pp[j + 3];
            rank += (u0 > v); rank += (u1 > v); rank += (u2 > v); rank += (u3 > v);
          }
          if (t < nP) ps[rank] = v;
        }
        NMX_WAVE_FENCE();
        NMX_TP(4)   // flush: sort
        // (2) merge into L_main in place.  Entry i moves to i + cnt(i), cnt(i) = pending samples > L[i]
        // (equal values: list entries first); pending sample j lands at ins[j] + j, ins[j] = list entries
        // >= ps[j].  Both lists are sorted, so one binary search per 64-entry BLOCK START (cb[b] = pending
        // samples > L[64 b]) confines everything else to the handful of pending samples between two block
        // starts.  The blocks are streamed from the last to the first: stores only go to addresses at or
        // above the block they come from, so the blocks loaded ahead (lower addresses) are never clobbered.
        if (nP > 0) {
          const int nblk = (Lm + 63) >> 6;
          for (int b = lane; b < nblk; b += 64) cb[b] = nmx_count_gt_lds(ps, nP, L[64 * b]);
          if (lane == 0) cb[nblk] = nP;
          NMX_WAVE_FENCE();
          if (LL) {
            // list in LDS: every pending sample finds its place by a binary search of its own while the list is still
            // whole (13 dependent LDS reads for 64 samples at a time), and below every list entry counts the pending
            // samples above it by a search between its block's two counters (2 - 3 steps) -- the walk over a block's
            // pending samples one by one (a broadcast read, a compare and a ballot each) was ~390 cycles per pending
            // sample: two thirds of a young stream's flush
            for (int j0 = 0; j0 < nP; j0 += 64) {
              const int j = j0 + lane;
              if (j < nP) ins[j] = nmx_count_ge_lds(L, Lm, ps[j]);
            }
            NMX_WAVE_FENCE();
          } else {
            for (int j = lane; j < cb[0]; j += 64) ins[j] = 0;
          }
          // NB blocks per step, and the loads of the NEXT step are issued before this step's entries are
          // placed (they lie at lower addresses than anything stored so far)
          constexpr int NB = 8;
          float vn[NB];
#pragma unroll
          for (int q = 0; q < NB; ++q) {
            const int i = 64 * (nblk - 1 - q) + lane;
            vn[q] = (nblk - 1 - q >= 0 && i < Lm) ? L[i] : 0.f;
          }
          for (int bb = nblk - 1; bb >= 0; bb -= NB) {
            float v[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) v[q] = vn[q];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
              const int i = 64 * (bb - NB - q) + lane;
              vn[q] = (bb - NB - q >= 0) ? L[i] : 0.f;   // (blocks below the top one are full)
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
              const int b = bb - q;
              if (b < 0) break;
              const int i = 64 * b + lane;
              const bool ok = i < Lm;
              const int c_lo = cb[b], c_hi = cb[b + 1];
              if (LL) {
                // pending samples > v[q]: all of [0, c_lo), none of [c_hi, nP).  (Measured and slower: the eight blocks'
                // searches in lockstep -- it pays the widest range eight times --, and reading the step's counters and
                // the first two pending samples of every range up front.)
                int lo2 = c_lo, hi2 = c_hi;
                while (lo2 < hi2) {
                  const int mid = (lo2 + hi2) >> 1;
                  if (ps[mid] > v[q]) lo2 = mid + 1; else hi2 = mid;
                }
                if (ok && lo2) L[i + lo2] = v[q];
                continue;
              }
              int cnt = c_lo;
              for (int j = c_lo; j < c_hi; ++j) {
                const float pj = ps[j];
                cnt += (pj > v[q]);
                const int at = 64 * b + __popcll(__ballot(ok && v[q] >= pj));
                if (lane == 0) ins[j] = at;
              }
              if (ok && cnt) L[i + cnt] = v[q];
            }
          }
          NMX_WAVE_FENCE();
          for (int j = lane; j < nP; j += 64) L[ins[j] + j] = ps[j];
        }
        NMX_TP(5)   // flush: stream + scatter
        Lm += nP;
        for (int j = lane; j < nF; j += 64) L[Lm + j] = F[fh + nF - 1 - j];
        // (Lm + nF == K by construction)
        if (!LL) __threadfence();   // the re-cut below reads what this wave just stored
        else NMX_WAVE_FENCE();
        nF = TREFILL;
        fh = 0;
        Lm = K - nF;
        nP = 0;
        for (int j = lane; j < nF; j += 64) F[j] = L[K - 1 - j];
        NMX_WAVE_FENCE();
        T = F[0];
        Fmax = F[nF - 1];
#ifdef NMX_THRW_PROFILE
        ++n_flush;
#endif
        NMX_TP(6)   // flush: re-cut
      }
    }
  }
#ifdef NMX_THRW_PROFILE
  if (lane == 0 && (sidx == 0 || sidx == 311))
    printf("[thrw %lld] hops %d inserts %d flushes %d | cycles: top %lld classify %lld fringe %lld store %lld sort %lld stream %lld recut %lld\n",
           sidx, A.n_windows, n_ins, n_flush, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6]);
#endif
  if (LL) {   // the list goes back to the state array (every launch ends in a flush: L is complete)
    NMX_WAVE_FENCE();
    for (int j = lane; j < K; j += 64) Lg[j] = L[j];
  }
  if (lane == 0) {
    A.counts[2 * sidx] = total;
    A.counts[2 * sidx + 1] = nwin;
  }
}
#endif

// ---------------------------------------------------------------------------------------
struct NmxBurstStatArgs {
  const float* env;   // [n_windows][C][Bb][W]
  const float* thr;   // [n_windows][C][Bb]
  float* out;
  int n_outputs, n_windows, n_channels, n_bands, W;
  float sfreq, seg_s;
  unsigned out_mask;  // bit i: slot i of {duration_mean, duration_max, amplitude_mean,
                      //                   amplitude_max, burst_rate_per_s, in_burst}
  NmxCols cols;       // a = band, b = slot among the enabled ones
  int off_e, off_red, lds_floats;
};

#ifdef NMX_HOST_EMU
NMX_DEV double nmx_wave_excl_sum_d(double v, double* total) { *total = v; return 0.0; }
// exclusive "latest" scan: carries (pos, val) of the lane with the largest pos before me
NMX_DEV void nmx_wave_excl_latest(int& pos, double& val) { pos = -1; val = 0.0; }
#else
// DPP scans (no LDS round trips): Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then
// row_bcast:15 / row_bcast:31 carry the row aggregates forward; a double moves as two dwords
NMX_DEV double nmx_dpp_d(double v, double ident, int ctrl_sel) {
  const long long vb = __builtin_bit_cast(long long, v), ib = __builtin_bit_cast(long long, ident);
  int lo = (int)(vb & 0xffffffffll), hi = (int)(vb >> 32);
  const int ilo = (int)(ib & 0xffffffffll), ihi = (int)(ib >> 32);
  switch (ctrl_sel) {   // compile-time after inlining: DPP controls are instruction immediates
    case 0: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x111, 0xf, 0xf, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x112, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x114, 0xf, 0xf, false); break;
    case 3: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x118, 0xf, 0xf, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x142, 0xa, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x142, 0xa, 0xf, false); break;
    case 5: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x143, 0xc, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x143, 0xc, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x138, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x138, 0xf, 0xf, false); break;  // wave_shr:1
  }
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
NMX_DEV int nmx_dpp_i(int v, int ident, int ctrl_sel) {
  switch (ctrl_sel) {
    case 0: return __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false);
    case 1: return __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false);
    case 2: return __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false);
    case 3: return __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false);
    case 4: return __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false);
    case 5: return __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false);
    default: return __builtin_amdgcn_update_dpp(ident, v, 0x138, 0xf, 0xf, false);
  }
}
NMX_DEV double nmx_wave_excl_sum_d(double v, double* total) {
  double inc = v;
#pragma unroll
  for (int st = 0; st < 6; ++st) inc += nmx_dpp_d(inc, 0.0, st);
  const long long b = __builtin_bit_cast(long long, inc);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  *total = __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
  return inc - v;
}
NMX_DEV void nmx_wave_excl_latest(int& pos, double& val) {
  int p = pos;
  double v = val;
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    const int tp = nmx_dpp_i(p, -1, st);
    const double tv = nmx_dpp_d(v, 0.0, st);
    if (tp > p) { p = tp; v = tv; }
  }
  // shift to exclusive (wave_shr:1; lane 0 reads the identity)
  pos = nmx_dpp_i(p, -1, 6);
  val = nmx_dpp_d(v, 0.0, 6);
}
#endif

#define NMX_EP(i) ((i) + ((i) >> 4))

// one WAVE (64 threads) per (window, channel, band)
NMX_DEV void nmx_burst_stat_item(const NmxBurstStatArgs& A, int w, int c, int bi, float* smem) {
  float* e = smem + A.off_e;
  float* red = smem + A.off_red;
  const int W = A.W;
  const long long item = ((long long)w * A.n_channels + c) * A.n_bands + bi;
  const float* src = A.env + item * W;
  // padded layout (one pad dword per 16 samples): per-lane contiguous chunks would otherwise hit
  // two LDS banks per half-wave (71 % of this kernel's LDS cycles were bank conflicts)
  nmx_stage_row(src, W, [=](int i, float v) { e[NMX_EP(i)] = v; });
  NMX_SYNC();
  const float thr = A.thr[item];
  // contiguous chunk per lane
  const int chunk = (W + NMX_NT - 1) / NMX_NT;
  const int i0 = NMX_TID * chunk, i1 = (i0 + chunk) < W ? (i0 + chunk) : W;
  // pass 1: chunk sum, last below-threshold position in chunk and prefix there
  double csum = 0.0;
  int zpos = -1;
  double zpre = 0.0;
  for (int i = i0; i < i1; ++i) {
    const float ev = e[NMX_EP(i)];
    csum += (double)ev;
    if (!(ev >= thr)) { zpos = i; zpre = csum; }
  }
  double total;
  const double base = nmx_wave_excl_sum_d(csum, &total);
  int carry_z = zpos;
  double carry_p = zpre + base;  // prefix (inclusive) at my last zero, global
  if (zpos < 0) carry_p = 0.0;
  nmx_wave_excl_latest(carry_z, carry_p);  // latest zero before my chunk
  // pass 2
  int n_above = 0, n_trans = 0, n_valid = 0, max_len = 0;
  double sum_means = 0.0;
  float amax = 0.f;
  double pre = base;
  int lastz = carry_z;
  double lastz_pre = carry_p;
  bool prev = (i0 > 0 && i0 < W) ? (e[NMX_EP(i0 - 1)] >= thr) : false;
  for (int i = i0; i < i1; ++i) {
    const float v = e[NMX_EP(i)];
    const bool b = v >= thr;
    if (b != prev) ++n_trans;
    if (b) {
      ++n_above;
      amax = v > amax ? v : amax;
    } else {
      if (prev) {  // a run [lastz + 1, i - 1] just finished -> valid run
        const int len = i - 1 - lastz;
        const double rs = pre - lastz_pre;
        sum_means += rs / (double)len;
        ++n_valid;
        max_len = len > max_len ? len : max_len;
      }
      lastz = i;
    }
    pre += (double)v;
    if (!b) lastz_pre = pre;
    prev = b;
  }
  n_above = nmx_block_sum_i(n_above, red);
  n_trans = nmx_block_sum_i(n_trans, red);
  n_valid = nmx_block_sum_i(n_valid, red);
  max_len = (int)nmx_block_max((float)max_len, red);
  amax = nmx_block_max(amax, red);
  // double sum via two floats would lose precision: reduce hi/lo parts
  const float hi = (float)sum_means;
  const float lo = (float)(sum_means - (double)hi);
  const double sm = (double)nmx_block_sum(hi, red) + (double)nmx_block_sum(lo, red);
  if (NMX_TID == 0) {
    const int num_bursts = n_trans / 2;
    const float dmean = num_bursts ? ((float)n_above / (float)num_bursts) / A.sfreq : 0.f;
    float vals[6];
    vals[0] = dmean;
    vals[1] = (float)max_len / A.sfreq;
    vals[2] = n_valid ? (float)(sm / (double)n_valid) : 0.f;
    vals[3] = amax;
    vals[4] = dmean / A.seg_s;
    vals[5] = (e[NMX_EP(W - 1)] >= thr) ? 1.f : 0.f;
    float* row = A.out + (long long)w * A.n_outputs;
    int col = A.cols.base + c * A.cols.ch_stride + bi * A.cols.a_stride;
    for (int s = 0; s < 6; ++s)
      if (A.out_mask & (1u << s)) {
        row[col] = vals[s];
        col += A.cols.b_stride;
      }
  }
}

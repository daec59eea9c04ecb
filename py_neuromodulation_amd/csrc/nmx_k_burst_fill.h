// nmx_k_burst_fill.h -- the threshold walk of a FRESH stream as one batch computation.
//
// While the percentile history fills (the first 291 hops at the default settings) every hop's threshold is an order
// statistic of a growing prefix of the sample sequence (nmx_k_bursts.h: thr = lerp(desc[m - 1 - lo], desc[m - 2 - lo])
// over ALL samples appended so far).  The per-hop merge of the workgroup kernel spends ~24 us per hop there (barriers,
// a 7 500-entry list shifted in LDS); here ONE workgroup per (channel, band)
//   1. loads every sample the first n hops append (W for the first hop, `overlap` for each later one) and sorts
//      them ONCE, descending, in LDS (bitonic, 1024 threads);
//   2. looks every sample up in the sorted array, all threads in parallel (binary search; equal values claim the next
//      free slot of their run with an atomic OR on a bit mask) and notes its slot, in arrival order, in a 16-bit
//      scratch list (global memory, L2-resident);
//   3. walks the hops in order on ONE wave, barrier-free: a hop's slots (staged in LDS 81 hops at a time) are marked
//      "arrived" in a second bit mask; the slot of the threshold's rank is a POINTER that moves by the few arrived
//      slots the rank changed by (one 4096-slot window of the mask per move: see "the rank pointer" below);
//   4. leaves the state the other walk kernels continue from: the descending top-K list and the counters.
// Same float64 interpolation on the same two floats as nmx_burst_thr_item: bit-identical thresholds.
#pragma once

#include "nmx_k_bursts.h"

#define NMX_FILL_NT 1024
#define NMX_FILL_CHUNK 8192  // slots staged in LDS at a time
#define NMX_FILL_MAX 32768   // samples (LDS: 4 bytes each + 2 bits + the slot chunk = 152 KB)

// LDS bytes for a sort size of n2 samples
static inline size_t nmx_burst_fill_lds(int n2) { return (size_t)n2 * 4 + 2 * ((size_t)n2 / 8) + 2 * NMX_FILL_CHUNK + 64; }
// The same computation as TWO launches (round 6, the default): steps 1 - 2 and the top-K head by a 1024-thread workgroup
// that holds the 128 KB sort (nmx_burst_fill_sort_item: ~0.1 ms, the only part that needs a CU to itself), step 3 by ONE
// wave per sequence with 21 KB of LDS (nmx_burst_fill_walk_item: the arrival mask, the staged slots and the two slots
// (rank, rank + 1) of every hop; the sorted values stay in global memory and are read once per hop, all hops in parallel,
// AFTER the walk).  The monolithic kernel kept 152 KB of a CU's LDS for the ~0.9 ms its one walking wave needs: 768
// sequences = three rounds over the 256 CUs, 3.1 ms during which the next chunk's filters found no CU to run on.
static inline size_t nmx_burst_fill_sort_lds(int n2) { return (size_t)n2 * 4 + (size_t)n2 / 8 + 64; }
static inline size_t nmx_burst_fill_walk_lds(int n2, int n_hops) {
  return (size_t)n2 / 8 + 2 * NMX_FILL_CHUNK + 4 * (size_t)n_hops + 64;
}

// hops a fresh stream can hand to this kernel: all samples of the batch must fit the LDS sort
static inline int nmx_burst_fill_hops(const NmxBurstThrArgs& A, int n_windows) {
  if (A.W > NMX_FILL_MAX || A.overlap < 1 || A.W < 2) return 0;
  const long long maxn = ((long long)NMX_FILL_MAX - A.W) / A.overlap + 1;
  return (int)(n_windows < maxn ? n_windows : maxn);
}

#ifndef NMX_HOST_EMU
// ---- the rank pointer -----------------------------------------------------------------------------------------
// The threshold of a hop is the arrived slot of rank r (r grows by ~(1 - q) x overlap per hop).  Instead of
// re-selecting it from counters, the walk keeps the slot p of the previous rank and MOVES it: the new arrivals below p
// raise p's own rank by their number (a ballot), the rest of the difference is a walk over the arrival mask -- ONE
// window of 64 x 64 slots read by the 64 lanes, a DPP scan of the popcounts and one popcount bisection.
NMX_DEV int nmx_fill_scan64(int v) {   // inclusive wave scan on the DPP path (no LDS round trips)
  int incl = v;
#define NMX_FILL_DPP(ctrl, rmask) __builtin_amdgcn_update_dpp(0, incl, ctrl, rmask, 0xf, false)
  incl += NMX_FILL_DPP(0x111, 0xf);   // row_shr:1
  incl += NMX_FILL_DPP(0x112, 0xf);   // row_shr:2
  incl += NMX_FILL_DPP(0x114, 0xf);   // row_shr:4
  incl += NMX_FILL_DPP(0x118, 0xf);   // row_shr:8
  incl += NMX_FILL_DPP(0x142, 0xa);   // row_bcast:15
  incl += NMX_FILL_DPP(0x143, 0xc);   // row_bcast:31
#undef NMX_FILL_DPP
  return incl;
}
// position of the (j + 1)-th lowest set bit of m (j < popcount(m))
NMX_DEV int nmx_fill_bit(unsigned long long m, int j) {
  unsigned x = (unsigned)m;
  int off = 0, cnt = __popc(x);
  if (j >= cnt) { j -= cnt; off = 32; x = (unsigned)(m >> 32); }
#pragma unroll
  for (int w = 16; w >= 1; w >>= 1) {
    cnt = __popc(x & ((1u << w) - 1u));
    if (j >= cnt) { j -= cnt; x >>= w; off += w; }
  }
  return off;
}
NMX_DEV unsigned long long nmx_fill_word(const unsigned* act, int w, int nw64) {
  return (w >= 0 && w < nw64) ? ((unsigned long long)act[2 * w] | ((unsigned long long)act[2 * w + 1] << 32)) : 0ull;
}
// the k-th arrived slot strictly AFTER p (k >= 1; p = -1: from the start); wave-uniform
NMX_DEV int nmx_fill_forward(const unsigned* act, int p, int k, int lane, int nw64) {
  int w0 = p < 0 ? 0 : (p >> 6);
  bool first = p >= 0;
  for (;;) {
    const int w = w0 + lane;
    unsigned long long m = nmx_fill_word(act, w, nw64);
    if (first && lane == 0) m &= ((p & 63) == 63) ? 0ull : ~((2ull << (p & 63)) - 1ull);   // bits <= p off
    const int c = __popcll(m), incl = nmx_fill_scan64(c);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total >= k) {
      const int src = (int)__ffsll((long long)__ballot(incl >= k)) - 1;
      int r = 0;
      if (lane == src) r = 64 * w + nmx_fill_bit(m, k - (incl - c) - 1);
      return __builtin_amdgcn_readlane(r, src);
    }
    k -= total;
    w0 += 64;
    first = false;
    if (w0 >= nw64) return 64 * nw64 - 1;   // (cannot happen: k never exceeds the arrived slots after p)
  }
}
// the k-th arrived slot strictly BEFORE p (k >= 1); wave-uniform
NMX_DEV int nmx_fill_backward(const unsigned* act, int p, int k, int lane, int nw64) {
  int w0 = p >> 6;
  bool first = true;
  for (;;) {
    const int w = w0 - lane;
    unsigned long long m = nmx_fill_word(act, w, nw64);
    if (first && lane == 0) m &= (1ull << (p & 63)) - 1ull;   // bits >= p off
    const int c = __popcll(m), incl = nmx_fill_scan64(c);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total >= k) {
      const int src = (int)__ffsll((long long)__ballot(incl >= k)) - 1;
      int r = 0;
      if (lane == src) r = 64 * w + 63 - nmx_fill_bit(__brevll(m), k - (incl - c) - 1);   // counted from the top
      return __builtin_amdgcn_readlane(r, src);
    }
    k -= total;
    w0 -= 64;
    first = false;
    if (w0 < 0) return 0;   // (cannot happen)
  }
}

NMX_DEV void nmx_burst_fill_item(const NmxBurstThrArgs& A, int c, int bi, int n2, unsigned short* slots, float* smem) {
  float* S = smem;                               // [n2] all samples, descending after the sort
  unsigned* act = (unsigned*)(S + n2);           // [n2 / 32] arrived bits
  unsigned* claim = act + n2 / 32;               // [n2 / 32] slots taken by the look-up
  unsigned short* sq = (unsigned short*)(claim + n2 / 32);   // [NMX_FILL_CHUNK] slots of the next hops
  const int tid = (int)threadIdx.x, nt = NMX_FILL_NT;
  const int W = A.W, ov = A.overlap, n = A.n_windows;
  const int M = W + (n - 1) * ov;
  const long long row = (long long)A.n_channels * A.n_bands * W;
  const float* e0 = A.env + ((long long)c * A.n_bands + bi) * W;
#ifdef NMX_FILL_PROFILE
  long long tp0 = clock64(), tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;
#endif
  for (int i = tid; i < n2; i += nt) {
    float v = -INFINITY;
    if (i < W) v = e0[i];
    else if (i < M) { const int h = 1 + (i - W) / ov, o = (i - W) - (h - 1) * ov; v = e0[(long long)h * row + (W - ov) + o]; }
    S[i] = v;
  }
  for (int i = tid; i < n2 / 16; i += nt) act[i] = 0u;   // (act and claim)
  __syncthreads();
#ifdef NMX_FILL_PROFILE
  tp1 = clock64();
#endif
  // Bitonic sort, one compare-exchange per thread and step: pair (i, i + j).  A wave's 64 consecutive pairs of a step
  // with j <= 64 lie inside ONE 128-element group, the same group for every such step: those steps need no workgroup
  // barrier (the LDS operations of one wave complete in order), only the steps with j >= 128 exchange between waves.
  for (int k = 2; k <= n2; k <<= 1) {
    int j = k >> 1;
    for (; j >= 128; j >>= 1) {
      for (int t = tid; t < n2 / 2; t += nt) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
        const float a = S[i], b = S[l];
        if (((i & k) == 0) ? (a < b) : (a > b)) { S[i] = b; S[l] = a; }
      }
      __syncthreads();
    }
    for (int t = tid; t < n2 / 2; t += nt)        // this wave's group for pair block t: steps j = min(k / 2, 64) .. 1
      for (int jj = j; jj > 0; jj >>= 1) {
        const int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), l = i + jj;
        const float a = S[i], b = S[l];
        if (((i & k) == 0) ? (a < b) : (a > b)) { S[i] = b; S[l] = a; }
      }
    if (k >= 128) __syncthreads();
  }
  __syncthreads();
#ifdef NMX_FILL_PROFILE
  tp2 = clock64();
#endif
  // ---- slot of every sample, in arrival order (sample i of the load above) ----
  unsigned short* sl = slots + ((long long)c * A.n_bands + bi) * NMX_FILL_MAX;
  for (int i = tid; i < M; i += nt) {
    float x;
    if (i < W) x = e0[i];
    else { const int h = 1 + (i - W) / ov, o = (i - W) - (h - 1) * ov; x = e0[(long long)h * row + (W - ov) + o]; }
    int lo = 0, hi = M;                         // first slot with S[slot] <= x (x is one of the sorted values)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (S[mid] > x) lo = mid + 1; else hi = mid; }
    int p = lo;
    for (;;) {                                  // equal values: the next free slot of the run
      const unsigned bit = 1u << (p & 31);
      if (!(atomicOr(&claim[p >> 5], bit) & bit)) break;
      ++p;
    }
    sl[i] = (unsigned short)p;
  }
#ifdef NMX_FILL_PROFILE
  tp3 = clock64();
#endif
  __syncthreads();   // (the wave below reads slots other waves wrote: workgroup-scope release / acquire)
  if (tid < 64) {   // ---- the walk: one wave, no barriers (LDS operations of one wave complete in order) ----
    const int lane = tid;
    const long long sidx = (long long)c * A.n_bands + bi;
    const int nw64 = n2 / 64;
    long long total = 0;
    int pa = 0x7fffffff, ra_prev = 0;   // the rank pointer (hop 0: nothing is 'below' it)
    // the slots of the hops travel from the scratch list to LDS in chunks of `nh` hops (one burst of pipelined loads
    // every nh hops instead of one global-memory round trip per hop)
    const int nh = ov <= NMX_FILL_CHUNK ? NMX_FILL_CHUNK / ov : 0;
#ifdef NMX_FILL_PROFILE
    long long wa = 0, wb = 0, wc = 0, wl;
#endif
    for (int h = 0; h < n; ++h) {
#ifdef NMX_FILL_PROFILE
      wl = clock64();
#endif
      const int n_new = h ? ov : W;
      int below = 0;   // this hop's arrivals below the rank pointer
      if (h == 0 || nh == 0) {
        const int off = h ? W + (h - 1) * ov : 0;
        for (int t = lane; t < n_new; t += 64) {
          const int p = (int)sl[off + t];
          atomicOr(&act[p >> 5], 1u << (p & 31));
          below += __popcll(__ballot(p < pa));
        }
      } else {
        const int hc = (h - 1) % nh;
        if (hc == 0) {
          const int cnt = ((n - h) < nh ? (n - h) : nh) * ov;
          const unsigned short* g = sl + W + (h - 1) * ov;
          for (int t = lane; t < cnt; t += 64) sq[t] = g[t];
        }
        for (int t = lane; t < ov; t += 64) {
          const int p = (int)sq[hc * ov + t];
          atomicOr(&act[p >> 5], 1u << (p & 31));
          below += __popcll(__ballot(p < pa));
        }
      }
      total += n_new;
      below = __builtin_amdgcn_readfirstlane(below);   // (lane 0 took part in every iteration above)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#ifdef NMX_FILL_PROFILE
      { const long long t_ = clock64(); wa += t_ - wl; wl = t_; }
#endif
      const long long m = total < (long long)A.n_ring ? total : (long long)A.n_ring;
      const double pos = A.q * (double)(m - 1);
      const long long lo_q = (long long)floor(pos);
      const bool have_hi = lo_q + 1 <= m - 1;
#ifdef NMX_FILL_PROFILE
      { const long long t_ = clock64(); wb += t_ - wl; wl = t_; }
#endif
      const int ra = (int)(m - 1 - lo_q);
      // p held rank ra_prev; the arrivals below it moved it to rank ra_prev + below
      const int k = h ? ra - (ra_prev + below) : ra + 1;
      if (k > 0) pa = nmx_fill_forward(act, h ? pa : -1, k, lane, nw64);
      else if (k < 0) pa = nmx_fill_backward(act, pa, -k, lane, nw64);
      ra_prev = ra;
      const int pb = have_hi ? nmx_fill_backward(act, pa, 1, lane, nw64) : 0;
      if (lane == 0)
        A.thr[((long long)h * A.n_channels + c) * A.n_bands + bi] =
            nmx_lerp_thr((double)S[pa], have_hi ? (double)S[pb] : 0.0, pos - (double)lo_q, have_hi);
#ifdef NMX_FILL_PROFILE
      { const long long t_ = clock64(); wc += t_ - wl; wl = t_; }
#endif
    }
#ifdef NMX_FILL_PROFILE
    if (lane == 0 && blockIdx.x == 0) printf("walk ticks: activate %lld scan %lld select %lld\n", wa, wb, wc);
#endif
    if (lane == 0) { A.counts[2 * sidx] = total; A.counts[2 * sidx + 1] = (long long)n; }
  }
#ifdef NMX_FILL_PROFILE
  tp4 = clock64();
  if (tid == 0 && blockIdx.x == 0) printf("fill ticks: load %lld sort %lld lookup %lld walk %lld\n", tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
#endif
  __syncthreads();
  // every sample has arrived: the state is the head of the sorted array
  float* gtop = A.top + ((long long)c * A.n_bands + bi) * A.K;
  const int keep = M < A.K ? M : A.K;
  for (int i = tid; i < keep; i += nt) gtop[i] = S[i];
}

// ---- the two-launch form ------------------------------------------------------------------------------------------------
// steps 1 - 2 + the state's top-K head; `sorted` = [n_seq][NMX_FILL_MAX] floats in global memory
NMX_DEV void nmx_burst_fill_sort_item(const NmxBurstThrArgs& A, int c, int bi, int n2, unsigned short* slots, float* sorted,
                                      float* smem) {
  float* S = smem;                               // [n2] all samples, descending after the sort
  unsigned* claim = (unsigned*)(S + n2);         // [n2 / 32] slots taken by the look-up
  const int tid = (int)threadIdx.x, nt = NMX_FILL_NT;
  const int W = A.W, ov = A.overlap, n = A.n_windows;
  const int M = W + (n - 1) * ov;
  const long long row = (long long)A.n_channels * A.n_bands * W;
  const long long sidx = (long long)c * A.n_bands + bi;
  const float* e0 = A.env + sidx * W;
  for (int i = tid; i < n2; i += nt) {
    float v = -INFINITY;
    if (i < W) v = e0[i];
    else if (i < M) { const int h = 1 + (i - W) / ov, o = (i - W) - (h - 1) * ov; v = e0[(long long)h * row + (W - ov) + o]; }
    S[i] = v;
  }
  for (int i = tid; i < n2 / 32; i += nt) claim[i] = 0u;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {   // (the bitonic network of nmx_burst_fill_item)
    int j = k >> 1;
    for (; j >= 128; j >>= 1) {
      for (int t = tid; t < n2 / 2; t += nt) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
        const float a = S[i], b = S[l];
        if (((i & k) == 0) ? (a < b) : (a > b)) { S[i] = b; S[l] = a; }
      }
      __syncthreads();
    }
    for (int t = tid; t < n2 / 2; t += nt)
      for (int jj = j; jj > 0; jj >>= 1) {
        const int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), l = i + jj;
        const float a = S[i], b = S[l];
        if (((i & k) == 0) ? (a < b) : (a > b)) { S[i] = b; S[l] = a; }
      }
    if (k >= 128) __syncthreads();
  }
  __syncthreads();
  unsigned short* sl = slots + sidx * NMX_FILL_MAX;
  for (int i = tid; i < M; i += nt) {
    float x;
    if (i < W) x = e0[i];
    else { const int h = 1 + (i - W) / ov, o = (i - W) - (h - 1) * ov; x = e0[(long long)h * row + (W - ov) + o]; }
    int lo = 0, hi = M;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (S[mid] > x) lo = mid + 1; else hi = mid; }
    int p = lo;
    for (;;) {
      const unsigned bit = 1u << (p & 31);
      if (!(atomicOr(&claim[p >> 5], bit) & bit)) break;
      ++p;
    }
    sl[i] = (unsigned short)p;
  }
  float* Sg = sorted + sidx * NMX_FILL_MAX;
  for (int i = tid; i < M; i += nt) Sg[i] = S[i];
  float* gtop = A.top + sidx * A.K;
  const int keep = M < A.K ? M : A.K;
  for (int i = tid; i < keep; i += nt) gtop[i] = S[i];
}

// step 3: ONE wave per sequence (a 64-thread workgroup); the thresholds from the sorted values in global memory, every hop
// by its own lane, after the walk (same two floats, same float64 interpolation: bit-identical to nmx_burst_fill_item)
NMX_DEV void nmx_burst_fill_walk_item(const NmxBurstThrArgs& A, int c, int bi, int n2, const unsigned short* slots,
                                      const float* sorted, float* smem) {
  unsigned* act = (unsigned*)smem;                          // [n2 / 32] arrived bits
  unsigned short* sq = (unsigned short*)(act + n2 / 32);    // [NMX_FILL_CHUNK] slots of the next hops
  unsigned short* pp = sq + NMX_FILL_CHUNK;                 // [2 n] the slots of rank and rank + 1 at every hop
  const int lane = (int)threadIdx.x;
  const int W = A.W, ov = A.overlap, n = A.n_windows;
  const long long sidx = (long long)c * A.n_bands + bi;
  const unsigned short* sl = slots + sidx * NMX_FILL_MAX;
  const int nw64 = n2 / 64;
  for (int i = lane; i < n2 / 32; i += 64) act[i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  long long total = 0;
  int pa = 0x7fffffff, ra_prev = 0;
  const int nh = ov <= NMX_FILL_CHUNK ? NMX_FILL_CHUNK / ov : 0;
  for (int h = 0; h < n; ++h) {
    const int n_new = h ? ov : W;
    int below = 0;
    if (h == 0 || nh == 0) {
      const int off = h ? W + (h - 1) * ov : 0;
      for (int t = lane; t < n_new; t += 64) {
        const int p = (int)sl[off + t];
        atomicOr(&act[p >> 5], 1u << (p & 31));
        below += __popcll(__ballot(p < pa));
      }
    } else {
      const int hc = (h - 1) % nh;
      if (hc == 0) {
        const int cnt = ((n - h) < nh ? (n - h) : nh) * ov;
        const unsigned short* g = sl + W + (h - 1) * ov;
        for (int t = lane; t < cnt; t += 64) sq[t] = g[t];
      }
      for (int t = lane; t < ov; t += 64) {
        const int p = (int)sq[hc * ov + t];
        atomicOr(&act[p >> 5], 1u << (p & 31));
        below += __popcll(__ballot(p < pa));
      }
    }
    total += n_new;
    below = __builtin_amdgcn_readfirstlane(below);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const long long m = total < (long long)A.n_ring ? total : (long long)A.n_ring;
    const double pos = A.q * (double)(m - 1);
    const long long lo_q = (long long)floor(pos);
    const bool have_hi = lo_q + 1 <= m - 1;
    const int ra = (int)(m - 1 - lo_q);
    const int k = h ? ra - (ra_prev + below) : ra + 1;
    if (k > 0) pa = nmx_fill_forward(act, h ? pa : -1, k, lane, nw64);
    else if (k < 0) pa = nmx_fill_backward(act, pa, -k, lane, nw64);
    ra_prev = ra;
    const int pb = have_hi ? nmx_fill_backward(act, pa, 1, lane, nw64) : 0;
    if (lane == 0) { pp[2 * h] = (unsigned short)pa; pp[2 * h + 1] = (unsigned short)pb; }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const float* Sg = sorted + sidx * NMX_FILL_MAX;
  for (int h = lane; h < n; h += 64) {
    const long long tot = (long long)W + (long long)h * ov;
    const long long m = tot < (long long)A.n_ring ? tot : (long long)A.n_ring;
    const double pos = A.q * (double)(m - 1);
    const long long lo_q = (long long)floor(pos);
    const bool have_hi = lo_q + 1 <= m - 1;
    A.thr[((long long)h * A.n_channels + c) * A.n_bands + bi] =
        nmx_lerp_thr((double)Sg[pp[2 * h]], have_hi ? (double)Sg[pp[2 * h + 1]] : 0.0, pos - (double)lo_q, have_hi);
  }
  if (lane == 0) { A.counts[2 * sidx] = total; A.counts[2 * sidx + 1] = (long long)n; }
}
#endif

#ifdef NMX_HOST_EMU
// Single-thread form of the same walk for the CPU logic tests (tests/emu): sort once, slot look-up with the claim
// mask, arrival mask and the MOVING rank pointer -- the steps of nmx_burst_fill_item without the wave mechanics.
#include <algorithm>
#include <functional>
inline void nmx_burst_fill_item_emu(const NmxBurstThrArgs& A, int c, int bi) {
  const int W = A.W, ov = A.overlap, n = A.n_windows;
  const int M = W + (n - 1) * ov;
  const long long row = (long long)A.n_channels * A.n_bands * W;
  const float* e0 = A.env + ((long long)c * A.n_bands + bi) * W;
  std::vector<float> arr(M);
  for (int i = 0; i < M; ++i) {
    if (i < W) arr[i] = e0[i];
    else { const int h = 1 + (i - W) / ov, o = (i - W) - (h - 1) * ov; arr[i] = e0[(long long)h * row + (W - ov) + o]; }
  }
  std::vector<float> S(arr);
  std::sort(S.begin(), S.end(), std::greater<float>());
  std::vector<char> claim(M, 0), act(M, 0);
  std::vector<int> slot(M);
  for (int i = 0; i < M; ++i) {
    int lo = 0, hi = M;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (S[mid] > arr[i]) lo = mid + 1; else hi = mid; }
    while (claim[lo]) ++lo;
    claim[lo] = 1;
    slot[i] = lo;
  }
  auto forward = [&](int p, int k) { while (k > 0) { ++p; if (act[p]) --k; } return p; };
  auto backward = [&](int p, int k) { while (k > 0) { --p; if (act[p]) --k; } return p; };
  long long total = 0;
  int pa = 0x7fffffff, ra_prev = 0;
  for (int h = 0; h < n; ++h) {
    const int n_new = h ? ov : W, off = h ? W + (h - 1) * ov : 0;
    int below = 0;
    for (int t = 0; t < n_new; ++t) { const int p = slot[off + t]; act[p] = 1; below += p < pa; }
    total += n_new;
    const long long m = total < (long long)A.n_ring ? total : (long long)A.n_ring;
    const double pos = A.q * (double)(m - 1);
    const long long lo_q = (long long)floor(pos);
    const bool have_hi = lo_q + 1 <= m - 1;
    const int ra = (int)(m - 1 - lo_q);
    const int k = h ? ra - (ra_prev + below) : ra + 1;
    if (k > 0) pa = forward(h ? pa : -1, k);
    else if (k < 0) pa = backward(pa, -k);
    ra_prev = ra;
    const int pb = have_hi ? backward(pa, 1) : 0;
    A.thr[((long long)h * A.n_channels + c) * A.n_bands + bi] =
        nmx_lerp_thr((double)S[pa], have_hi ? (double)S[pb] : 0.0, pos - (double)lo_q, have_hi);
  }
  const long long sidx = (long long)c * A.n_bands + bi;
  A.counts[2 * sidx] = total;
  A.counts[2 * sidx + 1] = (long long)n;
  float* gtop = A.top + sidx * A.K;
  const int keep = M < A.K ? M : A.K;
  for (int i = 0; i < keep; ++i) gtop[i] = S[i];
}
#endif

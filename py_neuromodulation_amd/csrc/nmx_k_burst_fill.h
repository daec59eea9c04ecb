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
//   3. walks the hops in order on ONE wave, barrier-free: a hop's slots (prefetched one hop ahead) are marked
//      "arrived" in a second bit mask, per 64 slots a counter holds the arrived ones; the two order statistics are
//      two rank selections (wave scan over the block counters, then inside one 64-bit mask);
//   4. leaves the state the other walk kernels continue from: the descending top-K list and the counters.
// Same float64 interpolation on the same two floats as nmx_burst_thr_item: bit-identical thresholds.
#pragma once

#include "nmx_k_bursts.h"

#define NMX_FILL_NT 1024
#define NMX_FILL_CHUNK 8192  // slots staged in LDS at a time
#define NMX_FILL_MAX 32768   // samples (LDS: 4 bytes each + 2 bits + 1/16 counter byte + the slot chunk = 154 KB)

// LDS bytes for a sort size of n2 samples
static inline size_t nmx_burst_fill_lds(int n2) { return (size_t)n2 * 4 + 2 * ((size_t)n2 / 8) + (size_t)n2 / 16 + 2 * NMX_FILL_CHUNK + 64; }

// hops a fresh stream can hand to this kernel: all samples of the batch must fit the LDS sort
static inline int nmx_burst_fill_hops(const NmxBurstThrArgs& A, int n_windows) {
  if (A.W > NMX_FILL_MAX || A.overlap < 1 || A.W < 2) return 0;
  const long long maxn = ((long long)NMX_FILL_MAX - A.W) / A.overlap + 1;
  return (int)(n_windows < maxn ? n_windows : maxn);
}

#ifndef NMX_HOST_EMU
// slots of the (r + 1)-th and, if r > 0, the r-th arrived sample in descending order (r < number arrived); wave-uniform.
// pre[8]: this lane's inclusive running counts over its `per` blocks, base: arrived before this lane's blocks.
// The r-th is the previous set bit of the same 64-slot mask when there is one (else *prev = -1: select it separately).
NMX_DEV int nmx_fill_select(const unsigned* act, const int* pre, int base, int lane_total, int per, int r, int lane, int* prev) {
  const bool mine = r >= base && r < base + lane_total;
  const int src = (int)__ffsll((long long)__ballot(mine)) - 1;
  int p = 0, q = -1;
  if (mine) {
    int rr = r - base, b = 0, skip = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // (compile-time indices only: a runtime index would put pre[] into scratch memory)
      const int before = i ? pre[i - 1] : 0;
      if (rr >= before && rr < pre[i]) { b = i; skip = before; }
    }
    rr -= skip;
    const int blk = per * lane + b;
    // the (rr + 1)-th set bit of the block's 64-bit mask (slots ascend = values descend): popcount bisection
    const unsigned long long m64 = (unsigned long long)act[2 * blk] | ((unsigned long long)act[2 * blk + 1] << 32);
    unsigned x = (unsigned)m64;
    int off = 0, cnt = __popc(x);
    if (rr >= cnt) { rr -= cnt; off = 32; x = (unsigned)(m64 >> 32); }
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
      cnt = __popc(x & ((1u << w) - 1u));
      if (rr >= cnt) { rr -= cnt; x >>= w; off += w; }
    }
    p = 64 * blk + off;
    const unsigned long long below = off ? (m64 & ((1ull << off) - 1ull)) : 0ull;
    if (below) q = 64 * blk + 63 - __clzll((long long)below);
  }
  *prev = __builtin_amdgcn_readlane(q, src);
  return __builtin_amdgcn_readlane(p, src);
}

NMX_DEV void nmx_burst_fill_item(const NmxBurstThrArgs& A, int c, int bi, int n2, unsigned short* slots, float* smem) {
  float* S = smem;                               // [n2] all samples, descending after the sort
  unsigned* act = (unsigned*)(S + n2);           // [n2 / 32] arrived bits
  unsigned* claim = act + n2 / 32;               // [n2 / 32] slots taken by the look-up
  int* cblk = (int*)(claim + n2 / 32);           // [n2 / 64] arrived per 64 slots
  unsigned short* sq = (unsigned short*)(cblk + n2 / 64);   // [NMX_FILL_CHUNK] slots of the next hops
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
  for (int i = tid; i < n2 / 64; i += nt) cblk[i] = 0;
  __syncthreads();
#ifdef NMX_FILL_PROFILE
  tp1 = clock64();
#endif
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < n2 / 2; t += nt) {   // one compare-exchange per thread and step: pair (i, i + j)
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
        const float a = S[i], b = S[l];
        if (((i & k) == 0) ? (a < b) : (a > b)) { S[i] = b; S[l] = a; }
      }
      __syncthreads();
    }
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
    const int nblk = n2 / 64, per = (nblk + 63) / 64;   // blocks per lane (<= 8)
    long long total = 0;
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
      if (h == 0 || nh == 0) {
        const int off = h ? W + (h - 1) * ov : 0;
        for (int t = lane; t < n_new; t += 64) {
          const int p = (int)sl[off + t];
          atomicOr(&act[p >> 5], 1u << (p & 31));
          atomicAdd(&cblk[p >> 6], 1);
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
          atomicAdd(&cblk[p >> 6], 1);
        }
      }
      total += n_new;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#ifdef NMX_FILL_PROFILE
      { const long long t_ = clock64(); wa += t_ - wl; wl = t_; }
#endif
      // running counts over this lane's blocks, exclusive scan of the lane totals
      int pre[8], tot = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = per * lane + i;
        tot += (i < per && b < nblk) ? cblk[b] : 0;
        pre[i] = tot;
      }
      int incl = tot;   // inclusive scan of the lane totals on the DPP path (no LDS round trips)
#define NMX_FILL_DPP(ctrl, rmask) __builtin_amdgcn_update_dpp(0, incl, ctrl, rmask, 0xf, false)
      incl += NMX_FILL_DPP(0x111, 0xf);   // row_shr:1
      incl += NMX_FILL_DPP(0x112, 0xf);   // row_shr:2
      incl += NMX_FILL_DPP(0x114, 0xf);   // row_shr:4
      incl += NMX_FILL_DPP(0x118, 0xf);   // row_shr:8
      incl += NMX_FILL_DPP(0x142, 0xa);   // row_bcast:15
      incl += NMX_FILL_DPP(0x143, 0xc);   // row_bcast:31
#undef NMX_FILL_DPP
      const int base = incl - tot;
      const long long m = total < (long long)A.n_ring ? total : (long long)A.n_ring;
      const double pos = A.q * (double)(m - 1);
      const long long lo_q = (long long)floor(pos);
      const bool have_hi = lo_q + 1 <= m - 1;
#ifdef NMX_FILL_PROFILE
      { const long long t_ = clock64(); wb += t_ - wl; wl = t_; }
#endif
      const int ra = (int)(m - 1 - lo_q);
      int pb, dummy;
      const int pa = nmx_fill_select(act, pre, base, tot, per, ra, lane, &pb);
      if (have_hi && pb < 0) pb = nmx_fill_select(act, pre, base, tot, per, ra - 1, lane, &dummy);
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
#endif

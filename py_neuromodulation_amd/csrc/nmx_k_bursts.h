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

// One hop of the sequential threshold walk costs ~4 barriers: rank the new samples by counting
// (broadcast LDS reads, no dependent chain), one binary search per NEW sample into the list,
// then an in-place merge in which every thread stages a contiguous chunk of the list in
// registers and writes it back shifted (merge path) -- no second list buffer, so two
// workgroups of this kernel only hold ~70 KiB of a CU's LDS and other kernels can overlap.
template <int CH>
NMX_DEV void nmx_burst_thr_item(const NmxBurstThrArgs& A, int c, int bi, float* smem) {
  float* L = smem + A.off_l0;
  float* pc = smem + A.off_p;            // [P2] raw new samples
  float* ps = pc + A.P2;                 // [P2] sorted (descending) new samples
  int* ins = (int*)(ps + A.P2);          // [P2] insertion index of ps[j] into L
  const int K = A.K, W = A.W;
  const long long sidx = (long long)c * A.n_bands + bi;
  long long total = A.counts[2 * sidx];
  long long nwin = A.counts[2 * sidx + 1];
  int len = (int)(total < K ? total : K);
  float* gtop = A.top + sidx * K;
  for (int i = NMX_TID; i < len; i += NMX_NT) L[i] = gtop[i];
  const int chunk = (K + NMX_NT - 1) / NMX_NT;   // <= CH
  const int i0 = NMX_TID * chunk;
#ifdef NMX_HOST_EMU
  std::vector<float> vals_store(K > 0 ? K : 1);
  float* vals = vals_store.data();
  const int CHB = chunk;   // the emulator's single "thread" owns the whole list
#else
  float vals[CH];
  constexpr int CHB = CH;
#endif
  // software prefetch of the next hop's new samples (one per thread) hides the HBM latency
  const bool can_prefetch = A.overlap <= NMX_NT;
  float pre = 0.f;
  bool have_pre = false;
  for (int w = 0; w < A.n_windows; ++w) {
    const int n_new = (nwin == 0) ? W : A.overlap;
    int n4 = (n_new + 3) & ~3;   // pc is padded with -inf to a multiple of 4 (float4 reads)
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
    NMX_SYNC();
    // 1. rank by counting -> ps descending (ties keep input order); float4 broadcast reads
    for (int t = NMX_TID; t < n_new; t += NMX_NT) {
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
    NMX_SYNC();
    // 2. insertion index of each new sample (after all list entries >= it)
    for (int j = NMX_TID; j < n_new; j += NMX_NT) ins[j] = nmx_count_ge(L, len, ps[j]);
    // stage my chunk of the list in registers
    const int i1 = (i0 + chunk) < len ? (i0 + chunk) : len;
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
    for (int k = 0; k < CHB; ++k) {
      if (k < chunk && i0 + k < i1) vals[k] = L[i0 + k];
    }
    NMX_SYNC();
    // 3. shifted write-back (merge path): entry i moves down by the number of new samples > L[i]
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
    len = (len + n_new) < K ? (len + n_new) : K;
    total += n_new;
    nwin += 1;
    if (NMX_TID == 0) {
      const long long m = total < A.n_ring ? total : A.n_ring;
      const double pos = A.q * (double)(m - 1);
      const long long lo = (long long)floor(pos);
      const double frac = pos - (double)lo;
      const double a = (double)L[m - 1 - lo];
      double r = a;
      if (lo + 1 <= m - 1) {
        const double b = (double)L[m - 2 - lo];
        const double d = b - a;
        r = (frac >= 0.5) ? b - d * (1.0 - frac) : a + d * frac;  // NumPy _lerp
      }
      A.thr[((long long)w * A.n_channels + c) * A.n_bands + bi] = (float)r;
    }
    // (no barrier needed here: the next hop only reads L until its own barrier 2)
  }
  NMX_SYNC();
  for (int i = NMX_TID; i < len; i += NMX_NT) gtop[i] = L[i];
  if (NMX_TID == 0) {
    A.counts[2 * sidx] = total;
    A.counts[2 * sidx + 1] = nwin;
  }
}

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
NMX_DEV double nmx_wave_excl_sum_d(double v, double* total) {
  const int lane = threadIdx.x & 63;
  double inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const double t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  *total = __shfl(inc, 63);
  return inc - v;
}
NMX_DEV void nmx_wave_excl_latest(int& pos, double& val) {
  const int lane = threadIdx.x & 63;
  int p = pos;
  double v = val;
  for (int o = 1; o < 64; o <<= 1) {
    const int tp = __shfl_up(p, o);
    const double tv = __shfl_up(v, o);
    if (lane >= o && tp > p) { p = tp; v = tv; }
  }
  // shift to exclusive
  const int ep = __shfl_up(p, 1);
  const double ev = __shfl_up(v, 1);
  pos = lane == 0 ? -1 : ep;
  val = lane == 0 ? 0.0 : ev;
}
#endif

// one WAVE (64 threads) per (window, channel, band)
NMX_DEV void nmx_burst_stat_item(const NmxBurstStatArgs& A, int w, int c, int bi, float* smem) {
  float* e = smem + A.off_e;
  float* red = smem + A.off_red;
  const int W = A.W;
  const long long item = ((long long)w * A.n_channels + c) * A.n_bands + bi;
  const float* src = A.env + item * W;
  for (int i = NMX_TID; i < W; i += NMX_NT) e[i] = src[i];
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
    csum += (double)e[i];
    if (!(e[i] >= thr)) { zpos = i; zpre = csum; }
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
  bool prev = (i0 > 0 && i0 < W) ? (e[i0 - 1] >= thr) : false;
  for (int i = i0; i < i1; ++i) {
    const float v = e[i];
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
    vals[5] = (e[W - 1] >= thr) ? 1.f : 0.f;
    float* row = A.out + (long long)w * A.n_outputs;
    int col = A.cols.base + c * A.cols.ch_stride + bi * A.cols.a_stride;
    for (int s = 0; s < 6; ++s)
      if (A.out_mask & (1u << s)) {
        row[col] = vals[s];
        col += A.cols.b_stride;
      }
  }
}

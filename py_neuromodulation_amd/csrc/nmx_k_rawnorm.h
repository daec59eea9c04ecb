// nmx_k_rawnorm.h -- raw_normalization, the last pre-processor of the chain.
//
// Reference: processing/normalization.py:31-116 with type "raw" (RawNormalizer): per channel, the
// history holds the whole first window and then the last `add` = int(sfreq / feat_hz) samples of
// every later window; at hop k >= 1 the statistics run over the history INCLUDING the current tail,
// the window becomes (x - mean) / std (zscore, std 0 -> 1) or (x - mean) / mean, is clipped and
// nan_to_num'ed; afterwards the history keeps its last N - 1 samples, N = int(time_s * sfreq).
// The first window ever is returned unchanged.
//
// Two kernels:
//  nmx_rawnorm_stats_item: one WAVE per channel walks the hops of the batch in order with sliding
//      float64 sums over a ring of N - 1 + add samples (the tails are summed by the 64 lanes, one DPP
//      reduction per quantity) and writes (mean, scale) per (hop, channel); scale = 0 marks the
//      pass-through first window.  Sequential by definition of the history, tiny.
//  nmx_rawnorm_apply: elementwise over [hops][C][W], writes the normalised windows.
#pragma once

#include "nmx_device.h"
#include "nmx_k_power.h"

#define NMX_RAWNORM_MEAN 1
#define NMX_RAWNORM_ZSCORE 2
// order-statistic methods (nmx_rawnorm_order_item): the history of every channel is ALSO kept sorted
#define NMX_RAWNORM_MEDIAN 3          // (x - nanmedian) / nanmedian            normalization.py:155-157
#define NMX_RAWNORM_ZSCORE_MEDIAN 4   // (x - nanmedian) / nanstd               normalization.py:166-169
#define NMX_RAWNORM_ROBUST 5          // sklearn RobustScaler  fitted on the history every hop (:57-70,172-186)
#define NMX_RAWNORM_MINMAX 6          // sklearn MinMaxScaler
#define NMX_RAWNORM_QUANTILE 7        // sklearn QuantileTransformer(n_quantiles = 300): a 300-entry table per (hop, channel)
#define NMX_RAWNORM_POWER 8           // sklearn PowerTransformer (Yeo-Johnson): (lambda, mean, scale) per (hop, channel)
#define NMX_RAWNORM_NQ 300
#define NMX_RAWNORM_SUBSAMPLE 10000   // QuantileTransformer.subsample: longer histories are randomly subsampled

struct NmxRawNormArgs {
  const float* x;            // windows: stream + starts, or materialised [n][C][W]
  long long ch_stride, win_stride;
  const long long* starts;   // may be null
  float* y;                  // [n_windows][C][W]
  int n_windows, n_channels, W;
  int add;                   // samples appended per hop
  int keep;                  // N - 1: history length kept between hops
  int cap;                   // ring capacity >= max(W, keep) + add
  int method;
  int clean_on_load;         // nan_to_num the input (no earlier stage did)
  float clip;                // <= 0: none
  long long hop0;            // hops seen before this batch
  float* ring;               // [C][cap]
  long long* count;          // [C] samples appended so far (ring write position = count % cap)
  int* len;                  // [C] current history length
  float* mean;               // [n_windows][C]
  float* scale;              // [n_windows][C]  1 / std or 1 / mean; 0 = pass-through
  // order-statistic methods: the history sorted ascending, double-buffered [C][2][cap]; cur[c] = buffer in use;
  // sorted_valid = 0: rebuild it from the ring first (fresh plan, reset, imported state)
  float* sorted;
  int* cur;
  int sorted_valid;
  int max_list;              // list capacity: W + add (inserted / dropped values of one hop)
  float* lists;              // null: the six lists live in LDS; else [C][6 max_list] in device memory (long windows)
  // "quantile": qt[n_windows][C][300] the fitted quantiles, qn[n_windows][C] how many of them (min(300, history));
  // sub[C][10000] scratch of the random subsample, seed of its hash
  double* qt;
  int* qn;
  float* sub;
  unsigned seed;
  // "power": ring_sl[C][cap] = sign(x) log1p|x| of the ring's samples; pw[n_windows][C][3] = (lambda, mean, scale)
  double* ring_sl;
  double* pw;
};

#ifdef NMX_HOST_EMU
NMX_DEV double nmx_wave_sum_d(double v) { return v; }
#else
NMX_DEV double nmx_wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif

#ifdef NMX_HOST_EMU
NMX_DEV double nmx_wave_max_d(double v) { return v; }
#else
NMX_DEV double nmx_wave_max_d(double v) {
  for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}
#endif

// "power" on the raw history of one channel, the lanes of ONE WAVE sharing the history (nmx_k_power.h has the
// thread-serial form and the arithmetic: scipy's bounded Brent search of the Yeo-Johnson likelihood).  Every lane
// evaluates its stride of the history, the sums are wave reductions, so the search runs in lock step on all lanes.
struct NmxRnHist {
  const float* ring;
  const double* rsl;
  int cap, len;
  long long cnt;
  NMX_DEVM float x(int i) const { return ring[(cnt - len + i) % cap]; }
  NMX_DEVM double sl(int i) const { return rsl[(cnt - len + i) % cap]; }
};
NMX_DEV double nmx_rn_negllf(double lmb, const NmxRnHist& h, double sl_sum, double t0) {
  const double eps = 2.220446049250313e-16, tiny = 2.2250738585072014e-308;
  const bool l0 = fabs(lmb) < eps, l2 = !(fabs(lmb - 2.0) > eps);
  const double c2 = 2.0 - lmb;
  auto tr = [&](double s) {
    const double l = fabs(s);
    return s >= 0.0 ? (l0 ? l : expm1(lmb * l) / lmb) : (l2 ? -l : -expm1(c2 * l) / c2);
  };
  const double k = tr(t0);   // shift = the transformed first sample of the history
  double s1 = 0.0, s2 = 0.0;
  for (int i = NMX_TID; i < h.len; i += NMX_NT) {
    const double d = tr(h.sl(i)) - k;
    s1 += d; s2 += d * d;
  }
  s1 = nmx_wave_sum_d(s1); s2 = nmx_wave_sum_d(s2);
  double var = (s2 - s1 * s1 / (double)h.len) / (double)h.len;
  if (var < 0.0) var = 0.0;
  if (var < tiny) return INFINITY;
  const double llf = -(double)h.len / 2.0 * log(var) + (lmb - 1.0) * sl_sum;
  if (llf == INFINITY || llf == -INFINITY) return INFINITY;
  return -llf;
}
NMX_DEV void nmx_rn_power_fit(const NmxRnHist& h, double& lmb, double& mean_t, double& scale_t) {
  const int n = h.len;
  double s = 0.0, amax = 0.0, sl_sum = 0.0, n_neg = 0.0, n_zero = 0.0;
  for (int i = NMX_TID; i < n; i += NMX_NT) {
    const double x = nmx_pw_clean(h.x(i));
    s += x;
    const double ax = fabs(x);
    amax = ax > amax ? ax : amax;
    n_neg += x < 0.0; n_zero += x == 0.0;
    sl_sum += h.sl(i);
  }
  s = nmx_wave_sum_d(s); sl_sum = nmx_wave_sum_d(sl_sum); n_neg = nmx_wave_sum_d(n_neg); n_zero = nmx_wave_sum_d(n_zero);
  amax = nmx_wave_max_d(amax);
  const double mean = s / (double)n;
  double q = 0.0;
  for (int i = NMX_TID; i < n; i += NMX_NT) { const double d = nmx_pw_clean(h.x(i)) - mean; q += d * d; }
  q = nmx_wave_sum_d(q);
  if (nmx_pw_constant(q / (double)n, mean, n) || n_zero == (double)n) {
    lmb = 1.0;
  } else {
    const double log_eps = log(2.220446049250313e-16), log1p_max_x = log1p(20.0 * amax);
    double lb = (log(2.2250738585072014e-308) - log_eps) / 2.0 / log1p_max_x;
    double ub = (log(1.7976931348623157e308) + log_eps) / 2.0 / log1p_max_x;
    if (n_neg == (double)n) { const double t = lb; lb = 2.0 - ub; ub = 2.0 - t; }
    else if (n_neg > 0.0) { const double l2 = 2.0 - ub, u2 = 2.0 - lb; lb = l2 > lb ? l2 : lb; ub = u2 < ub ? u2 : ub; }
    const double t0 = h.sl(0);
    lmb = nmx_pw_fminbound_f([&](double l) { return nmx_rn_negllf(l, h, sl_sum, t0); }, lb, ub);
  }
  const double k = nmx_pw_transform(nmx_pw_clean(h.x(0)), lmb);
  double s1 = 0.0, s2 = 0.0;
  for (int i = NMX_TID; i < n; i += NMX_NT) {
    const double d = nmx_pw_transform(nmx_pw_clean(h.x(i)), lmb) - k;
    s1 += d; s2 += d * d;
  }
  s1 = nmx_wave_sum_d(s1); s2 = nmx_wave_sum_d(s2);
  mean_t = k + s1 / (double)n;
  double var_t = (s2 - s1 * s1 / (double)n) / (double)n;
  if (var_t < 0.0) var_t = 0.0;
  scale_t = nmx_pw_constant(var_t, mean_t, n) ? 1.0 : sqrt(var_t);
}

// one wave per channel
NMX_DEV void nmx_rawnorm_stats_item(const NmxRawNormArgs& A, int c) {
  float* ring = A.ring + (long long)c * A.cap;
  double* rsl = A.method == NMX_RAWNORM_POWER ? A.ring_sl + (long long)c * A.cap : nullptr;
  if (rsl && !A.sorted_valid) {   // fresh plan / reset / imported state: derive it from the kept history
    const long long c0 = A.count[c];
    const int l0 = A.len[c];
    for (int i = NMX_TID; i < l0; i += NMX_NT) rsl[(c0 - l0 + i) % A.cap] = nmx_pw_sl(ring[(c0 - l0 + i) % A.cap]);
    NMX_SYNC();
  }
  long long cnt = A.count[c];
  int len = A.len[c];
  // sums of the kept history (rebuilt once per batch)
  double s1 = 0.0, s2 = 0.0;
  for (int i = NMX_TID; i < len; i += NMX_NT) {
    const double v = (double)ring[(cnt - len + i) % A.cap];
    s1 += v; s2 += v * v;
  }
  s1 = nmx_wave_sum_d(s1); s2 = nmx_wave_sum_d(s2);
  for (int w = 0; w < A.n_windows; ++w) {
    const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? A.starts[w] : 0ll);
    const bool first = (A.hop0 + w) == 0;
    const int n_new = first ? A.W : A.add;
    const float* tail = src + (A.W - n_new);
    double a1 = 0.0, a2 = 0.0;
    for (int i = NMX_TID; i < n_new; i += NMX_NT) {
      const float v = A.clean_on_load ? nmx_clean(tail[i]) : tail[i];
      ring[(cnt + i) % A.cap] = v;
      if (rsl) rsl[(cnt + i) % A.cap] = nmx_pw_sl(v);
      a1 += (double)v; a2 += (double)v * (double)v;
    }
    s1 += nmx_wave_sum_d(a1); s2 += nmx_wave_sum_d(a2);
    cnt += n_new; len += n_new;
    NMX_SYNC();
    float mean = 0.f, scale = 0.f;
    if (!first && A.method == NMX_RAWNORM_POWER) {
      NmxRnHist h;
      h.ring = ring; h.rsl = rsl; h.cap = A.cap; h.len = len; h.cnt = cnt;
      double lmb, mt, st;
      nmx_rn_power_fit(h, lmb, mt, st);
      if (NMX_TID == 0) {
        double* pw = A.pw + ((long long)w * A.n_channels + c) * 3;
        pw[0] = lmb; pw[1] = mt; pw[2] = st;
      }
      scale = 1.f;   // (not the pass-through marker)
    } else if (!first) {
      const double m = s1 / (double)len;
      if (A.method == NMX_RAWNORM_MEAN) {
        scale = (float)(1.0 / m);
      } else {
        double var = s2 / (double)len - m * m;
        if (var < 1e-9 * m * m) {   // cancellation: two-pass over the ring (rare)
          double acc = 0.0;
          for (int i = NMX_TID; i < len; i += NMX_NT) {
            const double d = (double)ring[(cnt - len + i) % A.cap] - m;
            acc += d * d;
          }
          var = nmx_wave_sum_d(acc) / (double)len;
        }
        const double sd = var > 0.0 ? sqrt(var) : 0.0;
        scale = (float)(1.0 / (sd == 0.0 ? 1.0 : sd));
      }
      mean = (float)m;
      // the (x - mean) difference is formed in float64 by the apply kernel from this mean
    }
    if (NMX_TID == 0) {
      A.mean[(long long)w * A.n_channels + c] = mean;
      A.scale[(long long)w * A.n_channels + c] = first ? 0.f : scale;
    }
    // history keeps its last N - 1 samples (the first call returns before the trim, :94-98)
    const int drop = (!first && len > A.keep) ? len - A.keep : 0;
    if (drop > 0) {
      double d1 = 0.0, d2 = 0.0;
      for (int i = NMX_TID; i < drop; i += NMX_NT) {
        const double v = (double)ring[(cnt - len + i) % A.cap];
        d1 += v; d2 += v * v;
      }
      const double dd2 = nmx_wave_sum_d(d2);
      s1 -= nmx_wave_sum_d(d1); s2 -= dd2;
      len -= drop;
      if (!(dd2 <= 1e4 * s2)) {   // what left dwarfs what stays (an artefact leaving the history): rebuild the sums
        double r1 = 0.0, r2 = 0.0;
        for (int i = NMX_TID; i < len; i += NMX_NT) {
          const double v = (double)ring[(cnt - len + i) % A.cap];
          r1 += v; r2 += v * v;
        }
        s1 = nmx_wave_sum_d(r1); s2 = nmx_wave_sum_d(r2);
      }
    }
  }
  if (NMX_TID == 0) { A.count[c] = cnt; A.len[c] = len; }
}

// ---- order-statistic methods ---------------------------------------------------------------------------
// One WORKGROUP per channel walks the hops of the batch in order.  Next to the ring (time order: which samples
// leave) the history is kept SORTED in global memory; per hop ONE merge pass removes the samples that left after
// the previous hop and inserts this hop's new ones:
//   * the <= W + add inserted and dropped values are rank-sorted in LDS;
//   * a thread per dropped value finds its position in the sorted history (binary search; equal values: the j-th
//     of a run of equals takes the j-th slot of the run), a thread per new value its insertion point;
//   * every element then moves to  i - #{dropped before i} + #{inserted before i}  (two binary searches in LDS).
// The median / quartiles / extremes of the hop are reads of the merged array.  O(history) per hop -- the methods
// are not on by default -- but every access is a coalesced stream and 256 channels run side by side.
#define NMX_RAWNORM_ORDER_NT 1024
NMX_DEV double nmx_rawnorm_block_sum_d(double v, double* red) {
#ifdef NMX_HOST_EMU
  (void)red;
  return v;
#else
  red[NMX_TID] = v;
  __syncthreads();
  for (int o = NMX_NT >> 1; o > 0; o >>= 1) {
    if (NMX_TID < o) red[NMX_TID] += red[NMX_TID + o];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
#endif
}
// sort v[0..n) ascending into o[0..n) by ranks (ties keep their order); n <= a few hundred after the first hop
NMX_DEV void nmx_rawnorm_rank_sort(const float* v, float* o, int n) {
  for (int i = NMX_TID; i < n; i += NMX_NT) {
    const float x = v[i];
    int r = 0;
    for (int k = 0; k < n; ++k) r += (v[k] < x) || (v[k] == x && k < i);
    o[r] = x;
  }
}
NMX_DEV int nmx_rawnorm_lower(const float* S, int n, float x) {   // first index with S[i] >= x
  int lo = 0, hi = n;
  while (lo < hi) { const int m = (lo + hi) >> 1; if (S[m] < x) lo = m + 1; else hi = m; }
  return lo;
}
NMX_DEV int nmx_rawnorm_upper(const float* S, int n, float x) {   // first index with S[i] > x
  int lo = 0, hi = n;
  while (lo < hi) { const int m = (lo + hi) >> 1; if (S[m] <= x) lo = m + 1; else hi = m; }
  return lo;
}
NMX_DEV int nmx_rawnorm_count_lt(const int* P, int n, int i) {    // #{P[k] < i}, P ascending
  int lo = 0, hi = n;
  while (lo < hi) { const int m = (lo + hi) >> 1; if (P[m] < i) lo = m + 1; else hi = m; }
  return lo;
}
NMX_DEV int nmx_rawnorm_count_le(const int* P, int n, int i) {    // #{P[k] <= i}
  int lo = 0, hi = n;
  while (lo < hi) { const int m = (lo + hi) >> 1; if (P[m] <= i) lo = m + 1; else hi = m; }
  return lo;
}
// np.percentile (linear, NumPy's _lerp) of the sorted history
NMX_DEV double nmx_rawnorm_quantile(const float* S, int n, double q) {
  const double vi = (double)(n - 1) * q;
  if (vi >= (double)(n - 1)) return (double)S[n - 1];
  const double fl = floor(vi);
  const int lo = (int)fl;
  const double t = vi - fl, a = (double)S[lo], b = (double)S[lo + 1], d = b - a;
  return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

// A uniformly random m-subset (m = 10 000) of the sorted history, in sorted order: every element gets a 32-bit hash
// key of (seed, index); the m smallest keys are the subset (ties by index).  The m-th smallest key comes from a
// four-pass radix select (256-bin histograms in LDS), the subset from a block scan of per-thread counts.
// `scr`: >= 2 * NMX_NT + 264 ints of LDS scratch (the whole carve-up of the item: 6 max_list + 2 NT words).  Statistically this is scikit-learn's resample(replace=False) per
// column (scikit-learn >= 1.5 draws ONE row subset for all columns; the sorted copies here are per channel).
NMX_DEV unsigned nmx_rawnorm_hash(unsigned seed, unsigned i) {
  unsigned h = seed ^ (i * 0x9E3779B9u);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
NMX_DEV void nmx_rawnorm_subsample(const float* S, int n, float* sub, unsigned seed, int* scr) {
  const int m = NMX_RAWNORM_SUBSAMPLE;
  int* hist = scr;                 // [256] + [8] control words
  int* ctl = scr + 256;            // ctl[0] = key prefix, ctl[1] = keys still to take inside the prefix class
  int* cntA = scr + 264;           // [NT]
  int* cntB = cntA + NMX_NT;       // [NT]
  if (NMX_TID == 0) { ctl[0] = 0; ctl[1] = m; }
  NMX_SYNC();
  for (int pass = 3; pass >= 0; --pass) {
    for (int i = NMX_TID; i < 256; i += NMX_NT) hist[i] = 0;
    NMX_SYNC();
    const unsigned prefix = (unsigned)ctl[0];
    const int sh = 8 * pass;
    for (int i = NMX_TID; i < n; i += NMX_NT) {
      const unsigned k = nmx_rawnorm_hash(seed, (unsigned)i);
      if (pass == 3 || (k >> (sh + 8)) == (prefix >> (sh + 8))) {
#ifdef NMX_HOST_EMU
        ++hist[(k >> sh) & 255u];
#else
        atomicAdd(&hist[(k >> sh) & 255u], 1);
#endif
      }
    }
    NMX_SYNC();
    if (NMX_TID == 0) {
      int need = ctl[1], b = 0;
      while (b < 255 && hist[b] < need) { need -= hist[b]; ++b; }
      ctl[0] = (int)(prefix | ((unsigned)b << sh));
      ctl[1] = need;   // keys to take among those equal to the prefix so far
    }
    NMX_SYNC();
  }
  const unsigned T = (unsigned)ctl[0];
  const int need_eq = ctl[1];      // of the keys == T, the first need_eq (by index) belong to the subset
  // contiguous blocks per thread: counts of (key < T) and (key == T)
  const int B = (n + NMX_NT - 1) / NMX_NT, i0 = NMX_TID * B, i1 = i0 + B < n ? i0 + B : n;
  int lt = 0, eq = 0;
  for (int i = i0; i < i1; ++i) {
    const unsigned k = nmx_rawnorm_hash(seed, (unsigned)i);
    lt += k < T; eq += k == T;
  }
  cntA[NMX_TID] = lt; cntB[NMX_TID] = eq;
  NMX_SYNC();
#ifndef NMX_HOST_EMU
  for (int o = 1; o < NMX_NT; o <<= 1) {   // inclusive Hillis-Steele scans of both count arrays
    const int a = NMX_TID >= o ? cntA[NMX_TID - o] : 0, b = NMX_TID >= o ? cntB[NMX_TID - o] : 0;
    __syncthreads();
    cntA[NMX_TID] += a; cntB[NMX_TID] += b;
    __syncthreads();
  }
#endif
  int out_lt = cntA[NMX_TID] - lt, seen_eq = cntB[NMX_TID] - eq;   // exclusive prefixes
  for (int i = i0; i < i1; ++i) {
    const unsigned k = nmx_rawnorm_hash(seed, (unsigned)i);
    if (k < T) {
      const int eq_before = seen_eq < need_eq ? seen_eq : need_eq;
      sub[out_lt + eq_before] = S[i];
      ++out_lt;
    } else if (k == T) {
      if (seen_eq < need_eq) sub[out_lt + seen_eq] = S[i];
      ++seen_eq;
    }
  }
  NMX_SYNC();
}

NMX_DEV void nmx_rawnorm_order_item(const NmxRawNormArgs& A, int c, float* smem) {
  float* ring = A.ring + (long long)c * A.cap;
  float* Sbuf = A.sorted + (long long)c * 2 * A.cap;
  const int ML = A.max_list;
  // window + hop beyond 6484 samples: the lists do not fit 160 KiB of LDS and live in device memory (L2-resident,
  // 24 bytes per sample and channel); LDS then holds the reduction scratch and the subsample scratch only
  float* lists = A.lists ? A.lists + (long long)c * 6 * ML : smem;
  float* in_raw = lists;             // [ML] this hop's new samples, time order
  float* in_s = in_raw + ML;         // [ML] sorted
  float* dr_raw = in_s + ML;         // [ML] samples that left after the previous hop
  float* dr_s = dr_raw + ML;         // [ML] sorted
  int* P = (int*)(dr_s + ML);        // [ML] positions of the dropped values in the sorted history
  int* Q = P + ML;                   // [ML] insertion points of the new values
  double* red = A.lists ? (double*)smem : (double*)(Q + ML);   // [NT] reduction scratch (8-byte aligned: ML is a multiple of 2)
  int* sub_scr = A.lists ? (int*)(smem + 2 * NMX_NT) : (int*)smem;   // 2 NT + 264 ints (the merge lists are dead there)
  long long cnt = A.count[c];
  int len = A.len[c];
  int cur = A.cur[c];
  float* S = Sbuf + (long long)cur * A.cap;
  if (!A.sorted_valid) {   // rebuild: copy the kept history and sort it (bitonic, padded with +inf)
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    cur = 0;      // n2 < 2 len <= 2 cap: the padded sort runs over BOTH (contiguous) buffers, the result sits in buffer 0
    S = Sbuf;
    for (int i = NMX_TID; i < n2; i += NMX_NT) S[i] = i < len ? ring[(cnt - len + i) % A.cap] : INFINITY;
    NMX_SYNC();
    for (int k = 2; k <= n2; k <<= 1)
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        for (int i = NMX_TID; i < n2; i += NMX_NT) {
          const int l = i ^ jj;
          if (l > i) {
            const float a = S[i], b = S[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { S[i] = b; S[l] = a; }
          }
        }
        NMX_SYNC();
      }
  }
  // sliding float64 sums for the standard deviation (zscore-median)
  double s1 = 0.0, s2 = 0.0;
  if (A.method == NMX_RAWNORM_ZSCORE_MEDIAN) {
    for (int i = NMX_TID; i < len; i += NMX_NT) {
      const double v = (double)ring[(cnt - len + i) % A.cap];
      s1 += v; s2 += v * v;
    }
    s1 = nmx_rawnorm_block_sum_d(s1, red); s2 = nmx_rawnorm_block_sum_d(s2, red);
  }
  int n_sorted = len;   // entries of S
  int n_drop = 0;       // pending removals (values in dr_raw)
  for (int w = 0; w <= A.n_windows; ++w) {
    const bool flush = w == A.n_windows;   // after the last hop: only the pending removals
    const bool first = !flush && (A.hop0 + w) == 0;
    const int n_new = flush ? 0 : (first ? A.W : A.add);
    if (flush && n_drop == 0) break;
    if (!flush) {
      const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? A.starts[w] : 0ll);
      const float* tail = src + (A.W - n_new);
      double a1 = 0.0, a2 = 0.0;
      for (int i = NMX_TID; i < n_new; i += NMX_NT) {
        const float v = A.clean_on_load ? nmx_clean(tail[i]) : tail[i];
        ring[(cnt + i) % A.cap] = v;
        in_raw[i] = v;
        a1 += (double)v; a2 += (double)v * (double)v;
      }
      if (A.method == NMX_RAWNORM_ZSCORE_MEDIAN) { s1 += nmx_rawnorm_block_sum_d(a1, red); s2 += nmx_rawnorm_block_sum_d(a2, red); }
      cnt += n_new; len += n_new;
    }
    NMX_SYNC();
    // ---- one merge pass: S - dropped + new -> S' -----------------------------------------------------------
    nmx_rawnorm_rank_sort(in_raw, in_s, n_new);
    nmx_rawnorm_rank_sort(dr_raw, dr_s, n_drop);
    NMX_SYNC();
    for (int j = NMX_TID; j < n_drop; j += NMX_NT) {
      const float d = dr_s[j];
      int t = 0;
      while (j - t - 1 >= 0 && dr_s[j - t - 1] == d) ++t;   // the t-th of a run of equal dropped values
      P[j] = nmx_rawnorm_lower(S, n_sorted, d) + t;
    }
    for (int j = NMX_TID; j < n_new; j += NMX_NT) Q[j] = nmx_rawnorm_upper(S, n_sorted, in_s[j]);
    NMX_SYNC();
    float* S2 = Sbuf + (long long)(1 - cur) * A.cap;
    {   // a thread's indices only grow: #{P < i} and #{Q <= i} advance instead of being searched for; eight loads
        // are in flight per thread (the walk is sequential over hops: memory latency is the whole cost)
      int nd = 0, ni = 0;
      for (int i0 = NMX_TID; i0 < n_sorted; i0 += 8 * NMX_NT) {
        float v[8];
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NMX_NT; v[u] = i < n_sorted ? S[i] : 0.f; }
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * NMX_NT;
          if (i >= n_sorted) break;
          while (nd < n_drop && P[nd] < i) ++nd;
          while (ni < n_new && Q[ni] <= i) ++ni;
          if (nd < n_drop && P[nd] == i) continue;   // this element leaves
          S2[i - nd + ni] = v[u];
        }
      }
    }
    for (int j = NMX_TID; j < n_new; j += NMX_NT) S2[Q[j] - nmx_rawnorm_count_lt(P, n_drop, Q[j]) + j] = in_s[j];
    NMX_SYNC();
    n_sorted += n_new - n_drop;
    cur = 1 - cur;
    S = S2;
    n_drop = 0;
    if (flush) break;
    // ---- "quantile": QuantileTransformer(n_quantiles = 300) fitted on this hop's history --------------------
    if (A.method == NMX_RAWNORM_QUANTILE && !first) {
      const float* Qs = S;
      int nqs = n_sorted;
      if (n_sorted > NMX_RAWNORM_SUBSAMPLE) {   // scikit-learn draws 10 000 of the history's rows at random
        float* sub = A.sub + (long long)c * NMX_RAWNORM_SUBSAMPLE;
        nmx_rawnorm_subsample(S, n_sorted, sub, A.seed ^ (unsigned)((A.hop0 + w) * 2654435761u) ^ ((unsigned)c * 40503u), sub_scr);
        Qs = sub;
        nqs = NMX_RAWNORM_SUBSAMPLE;
      }
      const int nq = nqs < NMX_RAWNORM_NQ ? nqs : NMX_RAWNORM_NQ;
      double* qt = A.qt + ((long long)w * A.n_channels + c) * NMX_RAWNORM_NQ;
      for (int i = NMX_TID; i < nq; i += NMX_NT)   // references = linspace(0, 1, nq); np.nanpercentile(history, references * 100)
        qt[i] = nq == 1 ? (double)Qs[0] : nmx_rawnorm_quantile(Qs, nqs, i < nq - 1 ? (double)i * (1.0 / (double)(nq - 1)) : 1.0);
      if (NMX_TID == 0) A.qn[(long long)w * A.n_channels + c] = nq;
      NMX_SYNC();
    }
    // ---- statistics of the hop ------------------------------------------------------------------------------
    if (NMX_TID == 0) {
      float center = 0.f, scale = 0.f;
      if (!first && A.method == NMX_RAWNORM_QUANTILE) {
        scale = 1.f;   // (not the pass-through marker; the apply kernel reads the table)
      } else if (!first) {
        const int n = n_sorted;
        const double med = (n & 1) ? (double)S[n >> 1] : 0.5 * ((double)S[(n >> 1) - 1] + (double)S[n >> 1]);
        if (A.method == NMX_RAWNORM_MEDIAN) {
          center = (float)med; scale = (float)(1.0 / med);
        } else if (A.method == NMX_RAWNORM_ZSCORE_MEDIAN) {
          const double m = s1 / (double)len;
          double var = s2 / (double)len - m * m;
          if (var < 1e-9 * m * m) {   // cancellation (rare): two-pass over the sorted copy
            double acc = 0.0;
            for (int i = 0; i < n; ++i) { const double d = (double)S[i] - m; acc += d * d; }
            var = acc / (double)n;
          }
          const double sd = var > 0.0 ? sqrt(var) : 0.0;
          center = (float)med; scale = (float)(1.0 / (sd == 0.0 ? 1.0 : sd));
        } else if (A.method == NMX_RAWNORM_ROBUST) {
          double sc = nmx_rawnorm_quantile(S, n, 0.75) - nmx_rawnorm_quantile(S, n, 0.25);
          if (sc < 10.0 * 2.220446049250313e-16) sc = 1.0;
          center = (float)med; scale = (float)(1.0 / sc);
        } else {   // minmax: x * s + (0 - lo * s) = (x - lo) * s up to rounding
          double rng = (double)S[n - 1] - (double)S[0];
          if (rng < 10.0 * 2.220446049250313e-16) rng = 1.0;
          center = S[0]; scale = (float)(1.0 / rng);
        }
        if (scale == 0.f) scale = 1e-45f;   // (0 is the pass-through marker; a huge spread rounds to the smallest float)
      }
      A.mean[(long long)w * A.n_channels + c] = center;
      A.scale[(long long)w * A.n_channels + c] = first ? 0.f : scale;
    }
    // ---- the history keeps its last N - 1 samples: they leave the sorted copy with the next merge ------------
    const int drop = (!first && len > A.keep) ? len - A.keep : 0;
    if (drop > 0) {
      double d1 = 0.0, d2 = 0.0;
      for (int i = NMX_TID; i < drop; i += NMX_NT) {
        const float v = ring[(cnt - len + i) % A.cap];
        dr_raw[i] = v;
        d1 += (double)v; d2 += (double)v * (double)v;
      }
      len -= drop;
      n_drop = drop;
      if (A.method == NMX_RAWNORM_ZSCORE_MEDIAN) {
        const double dd2 = nmx_rawnorm_block_sum_d(d2, red);
        s1 -= nmx_rawnorm_block_sum_d(d1, red); s2 -= dd2;
        if (!(dd2 <= 1e4 * s2)) {   // (as in nmx_rawnorm_stats_item: rebuild when what left dwarfs what stays)
          double r1 = 0.0, r2 = 0.0;
          for (int i = NMX_TID; i < len; i += NMX_NT) {
            const double v = (double)ring[(cnt - len + i) % A.cap];
            r1 += v; r2 += v * v;
          }
          s1 = nmx_rawnorm_block_sum_d(r1, red); s2 = nmx_rawnorm_block_sum_d(r2, red);
        }
      }
    }
    NMX_SYNC();
  }
  if (NMX_TID == 0) { A.count[c] = cnt; A.len[c] = len; A.cur[c] = cur; }
}

// QuantileTransformer.transform (uniform output) of one value against the fitted table Q[0..nq), references
// R[i] = i / (nq - 1):  y = (interp(x, Q, R) - interp(-x, -Q[::-1], -R[::-1])) / 2, x == Q[0] -> 0, x == Q[-1] -> 1
NMX_DEV double nmx_rawnorm_qt_apply(const double* Q, int nq, double x) {
  if (x != x) return x;
  const double q0 = Q[0], q1 = Q[nq - 1];
  if (x == q0) return 0.0;
  if (x == q1) return 1.0;
  if (x < q0) return 0.0;
  if (x > q1) return 1.0;
  auto ref = [&](int i) { return (i < nq - 1) ? (double)i * (1.0 / (double)(nq - 1)) : 1.0; };
  int lo = 0, hi = nq - 1;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (Q[mid] <= x) lo = mid; else hi = mid; }
  const int a = lo;      // LAST index with Q[a] <= x
  lo = 0; hi = nq - 1;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (Q[mid] >= x) hi = mid; else lo = mid; }
  const int b = hi;      // FIRST index with Q[b] >= x
  double r1, r2;
  if (Q[a] == x) r1 = ref(a);
  else r1 = (ref(a + 1) - ref(a)) / (Q[a + 1] - Q[a]) * (x - Q[a]) + ref(a);
  if (Q[b] == x) r2 = -ref(b);
  else r2 = ((-ref(b - 1)) - (-ref(b))) / ((-Q[b - 1]) - (-Q[b])) * ((-x) - (-Q[b])) + (-ref(b));
  return 0.5 * (r1 - r2);
}

// element i of window (w, c)
NMX_DEV void nmx_rawnorm_apply(const NmxRawNormArgs& A, long long idx) {
  const long long per_w = (long long)A.n_channels * A.W;
  if (idx >= (long long)A.n_windows * per_w) return;
  const int w = (int)(idx / per_w);
  const int r = (int)(idx - (long long)w * per_w);
  const int c = r / A.W, i = r - c * A.W;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? A.starts[w] : 0ll);
  const float x = A.clean_on_load ? nmx_clean(src[i]) : src[i];
  const float sc = A.scale[(long long)w * A.n_channels + c];
  float out = x;
  if (sc != 0.f && A.method == NMX_RAWNORM_QUANTILE) {
    const long long wc = (long long)w * A.n_channels + c;
    double o = nmx_rawnorm_qt_apply(A.qt + wc * NMX_RAWNORM_NQ, A.qn[wc], (double)x);
    if (A.clip > 0.f) o = o < -(double)A.clip ? -(double)A.clip : (o > (double)A.clip ? (double)A.clip : o);
    out = nmx_clean((float)o);
  } else if (sc != 0.f && A.method == NMX_RAWNORM_POWER) {
    const double* pw = A.pw + ((long long)w * A.n_channels + c) * 3;
    double o = (nmx_pw_transform((double)x, pw[0]) - pw[1]) / pw[2];
    if (A.clip > 0.f) o = o < -(double)A.clip ? -(double)A.clip : (o > (double)A.clip ? (double)A.clip : o);
    out = nmx_clean((float)o);
  } else if (sc != 0.f) {
    out = (x - A.mean[(long long)w * A.n_channels + c]) * sc;
    if (A.clip > 0.f) out = out < -A.clip ? -A.clip : (out > A.clip ? A.clip : out);
    out = nmx_clean(out);
  }
  A.y[idx] = out;
}

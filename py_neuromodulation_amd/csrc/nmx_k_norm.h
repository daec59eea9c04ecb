// nmx_k_norm.h -- feature normalisation over a batch of hops.
//
// Reference: processing/normalization.py:93-111 (Normalizer.process, type "feature") with
// norm_mean (:150-152) and norm_zscore (:160-163): for every hop the statistics run over the
// history INCLUDING the current row; afterwards the history keeps its last N - 1 rows
// (N = normalization_time_s * sampling_rate_features_hz); the very first row ever seen is returned
// untouched (:94-98); then clip (:104-105) and nan_to_num (:109).  nan_mean / nan_std ignore NaNs.
//
// The reference is sequential over hops (one Python call per hop).  Columns are independent, so
// one THREAD owns one feature column and walks the batch with sliding float64 sums (sum, sum of
// squares, count of non-NaN) over a ring of the last N raw values per column: O(1) per value
// instead of the reference's O(N).  Features are fp32 numbers, so the float64 sums of <= N of them
// are exact or within a few ulp; if the variance is small against mean^2 (cancellation) the two-pass
// form is evaluated from the ring instead.  Lanes = consecutive columns: every row access is coalesced.
//
// +-inf features (log10 of a zero power: a flat or all-NaN channel) stay OUT of the sliding sums -- inf - inf
// when the value leaves the window would poison them for good -- and are counted instead: while the window
// holds one, np.mean is +-inf (or NaN) and np.std is NaN, so mean / zscore / zscore-median give NaN -> 0
// (normalization.py:109); the median itself stays finite and is evaluated as usual.  Once the value has left
// the window the sums are exactly what they would have been without it, like the reference's recomputation.
#pragma once

#include "nmx_device.h"

#define NMX_NORM_MEAN 0
#define NMX_NORM_ZSCORE 1
#define NMX_NORM_MEDIAN 2
#define NMX_NORM_ZSCORE_MEDIAN 3

struct NmxNormArgs {
  float* rows;                 // [n_rows][ld]  in place
  long long ld;
  int n_rows, n_cols;
  const unsigned char* colmask;  // optional [n_cols]: 0 = leave the column alone ("psd" keys)
  float* ring;                 // [cap][n_cols] raw history, slot = seq % cap
  int cap;                     // N (window incl. the current row)
  long long seq0;              // rows seen before this batch
  int method;
  float clip;                  // <= 0: none
  float* sorted;               // median methods: [cap][n_cols] scratch, the window's non-NaN values ascending
};

// The median methods (normalization.py:155-157,166-169: np.median / np.nanmedian over the history incl.
// the current row) keep the window's non-NaN values SORTED per column: a new value is inserted and the
// value that leaves the window removed by shifting (<= N moves, lanes = columns, so every move is a
// coalesced row access); the median is then one or two reads.  The sorted copy is scratch -- it is rebuilt
// from the ring at the start of every batch, like the sums.
NMX_DEV bool nmx_norm_finite(float v) { return v - v == 0.f; }   // false for NaN and +-inf

NMX_DEV int nmx_norm_lower(const float* S, int n_cols, int j, int n, float x) {   // first index with S[idx] >= x
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (S[(long long)mid * n_cols + j] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// S[k] for k in [a, b) moves one slot up (to k + 1), highest first; four independent loads per step
NMX_DEV void nmx_norm_shift_up(float* S, int n_cols, int j, int a, int b) {
  int k = b;
  for (; k - 4 >= a; k -= 4) {
    float* q = S + (long long)(k - 4) * n_cols + j;
    const float t0 = q[0], t1 = q[n_cols], t2 = q[2ll * n_cols], t3 = q[3ll * n_cols];
    q[n_cols] = t0; q[2ll * n_cols] = t1; q[3ll * n_cols] = t2; q[4ll * n_cols] = t3;
  }
  for (; k > a; --k) S[(long long)k * n_cols + j] = S[(long long)(k - 1) * n_cols + j];
}
// S[k] for k in (a, b] moves one slot down (to k - 1), lowest first
NMX_DEV void nmx_norm_shift_down(float* S, int n_cols, int j, int a, int b) {
  int k = a;
  for (; k + 4 <= b; k += 4) {
    float* q = S + (long long)k * n_cols + j;
    const float t1 = q[n_cols], t2 = q[2ll * n_cols], t3 = q[3ll * n_cols], t4 = q[4ll * n_cols];
    q[0] = t1; q[n_cols] = t2; q[2ll * n_cols] = t3; q[3ll * n_cols] = t4;
  }
  for (; k < b; ++k) S[(long long)k * n_cols + j] = S[(long long)(k + 1) * n_cols + j];
}
NMX_DEV void nmx_norm_insert(float* S, int n_cols, int j, int& n, float x) {
  const int p = nmx_norm_lower(S, n_cols, j, n, x);
  nmx_norm_shift_up(S, n_cols, j, p, n);
  S[(long long)p * n_cols + j] = x;
  ++n;
}
NMX_DEV void nmx_norm_remove(float* S, int n_cols, int j, int& n, float x) {   // x is in the list
  const int p = nmx_norm_lower(S, n_cols, j, n, x);
  nmx_norm_shift_down(S, n_cols, j, p, n - 1);
  --n;
}
// remove `o` (in the list) and insert `x` in one pass: only the entries between the two positions move
NMX_DEV void nmx_norm_replace(float* S, int n_cols, int j, int n, float o, float x) {
  const int po = nmx_norm_lower(S, n_cols, j, n, o);
  const int px = nmx_norm_lower(S, n_cols, j, n, x);   // position of x in the list that still holds o
  if (px > po) {   // x lands above o: entries (po, px) move down, x takes slot px - 1
    nmx_norm_shift_down(S, n_cols, j, po, px - 1);
    S[(long long)(px - 1) * n_cols + j] = x;
  } else {         // x lands at or below o: entries [px, po) move up, x takes slot px
    nmx_norm_shift_up(S, n_cols, j, px, po);
    S[(long long)px * n_cols + j] = x;
  }
}
NMX_DEV double nmx_norm_median(const float* S, int n_cols, int j, int n) {
  if (n == 0) return NAN;
  const double hi = (double)S[(long long)(n >> 1) * n_cols + j];
  return (n & 1) ? hi : 0.5 * ((double)S[(long long)((n >> 1) - 1) * n_cols + j] + hi);
}

NMX_DEV void nmx_norm_column(const NmxNormArgs& A, int j) {
  if (j >= A.n_cols) return;
  if (A.colmask && !A.colmask[j]) return;
  const int cap = A.cap;
  // rebuild the sums from the history (the last min(seq0, cap - 1) rows)
  const long long have = A.seq0 < (long long)(cap - 1) ? A.seq0 : (long long)(cap - 1);
  double s1 = 0.0, s2 = 0.0;
  int cnt = 0, ninf = 0, len = (int)have;   // cnt: finite values in the window, ninf: +-inf values
  for (long long q = A.seq0 - have; q < A.seq0; ++q) {
    const float h = A.ring[(q % cap) * A.n_cols + j];
    if (nmx_norm_finite(h)) { s1 += (double)h; s2 += (double)h * (double)h; ++cnt; }
    else if (h == h) ++ninf;
  }
  const bool med = A.method >= NMX_NORM_MEDIAN;
  int ns = 0;   // entries of the sorted copy (== cnt)
  if (med)
    for (long long q = A.seq0 - have; q < A.seq0; ++q) {
      const float h = A.ring[(q % cap) * A.n_cols + j];
      if (h == h) nmx_norm_insert(A.sorted, A.n_cols, j, ns, h);
    }
  bool pend = false;   // median methods: the value trimmed after the previous hop leaves the sorted copy
  float pend_val = 0.f;   //   together with the next insertion (nothing reads the copy in between)
  for (int r = 0; r < A.n_rows; ++r) {
    const long long q = A.seq0 + r;
    float* cell = A.rows + (long long)r * A.ld + j;
    const float x = *cell;
    if (len == cap) {  // cannot happen with the trim below; kept for safety
      const float o = A.ring[((q - cap) % cap) * A.n_cols + j];
      if (o == o) {
        if (nmx_norm_finite(o)) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; } else --ninf;
        if (med) {
          if (pend) { nmx_norm_remove(A.sorted, A.n_cols, j, ns, pend_val); pend = false; }
          nmx_norm_remove(A.sorted, A.n_cols, j, ns, o);
        }
      }
      --len;
    }
    A.ring[(q % cap) * A.n_cols + j] = x;
    if (nmx_norm_finite(x)) { s1 += (double)x; s2 += (double)x * (double)x; ++cnt; }
    else if (x == x) ++ninf;
    if (med) {
      if (x == x && pend) nmx_norm_replace(A.sorted, A.n_cols, j, ns, pend_val, x);
      else if (x == x) nmx_norm_insert(A.sorted, A.n_cols, j, ns, x);
      else if (pend) nmx_norm_remove(A.sorted, A.n_cols, j, ns, pend_val);
      pend = false;
    }
    ++len;
    if (q > 0) {  // the first row ever is returned as it came
      double out;
      if (cnt + ninf == 0 || (ninf > 0 && A.method != NMX_NORM_MEDIAN)) {
        out = NAN;   // empty window, or +-inf inside it: mean +-inf / NaN, std NaN (see the header)
      } else if (A.method == NMX_NORM_MEDIAN) {
        const double m = nmx_norm_median(A.sorted, A.n_cols, j, ns);
        out = ((double)x - m) / m;
      } else {
        const double mean = s1 / (double)cnt;
        if (A.method == NMX_NORM_MEAN) {
          out = ((double)x - mean) / mean;
        } else {
          double var = s2 / (double)cnt - mean * mean;
          if (var < 1e-9 * mean * mean) {  // cancellation: two-pass over the ring (rare)
            double acc = 0.0;
            for (long long t = q - len + 1; t <= q; ++t) {
              const float h = A.ring[(t % cap) * A.n_cols + j];
              if (h == h) { const double d = (double)h - mean; acc += d * d; }
            }
            var = acc / (double)cnt;
          }
          double sd = var > 0.0 ? sqrt(var) : 0.0;
          if (sd == 0.0) sd = 1.0;
          const double centre = A.method == NMX_NORM_ZSCORE_MEDIAN ? nmx_norm_median(A.sorted, A.n_cols, j, ns) : mean;
          out = ((double)x - centre) / sd;
        }
      }
      if (A.clip > 0.f) {  // ndarray.clip: NaN stays NaN
        if (out < -(double)A.clip) out = -(double)A.clip;
        if (out > (double)A.clip) out = (double)A.clip;
      }
      *cell = nmx_clean((float)out);
    }
    // history keeps its last N - 1 rows (normalization.py:107)
    if (len > cap - 1) {
      const float o = A.ring[((q - (cap - 1)) % cap) * A.n_cols + j];
      if (o == o) {
        if (nmx_norm_finite(o)) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; } else --ninf;
        if (med) { pend = true; pend_val = o; }
      }
      --len;
    }
  }
}

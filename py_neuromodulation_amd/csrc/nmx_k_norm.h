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
#pragma once

#include "nmx_device.h"

#define NMX_NORM_MEAN 0
#define NMX_NORM_ZSCORE 1

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
};

NMX_DEV void nmx_norm_column(const NmxNormArgs& A, int j) {
  if (j >= A.n_cols) return;
  if (A.colmask && !A.colmask[j]) return;
  const int cap = A.cap;
  // rebuild the sums from the history (the last min(seq0, cap - 1) rows)
  const long long have = A.seq0 < (long long)(cap - 1) ? A.seq0 : (long long)(cap - 1);
  double s1 = 0.0, s2 = 0.0;
  int cnt = 0, len = (int)have;
  for (long long q = A.seq0 - have; q < A.seq0; ++q) {
    const float h = A.ring[(q % cap) * A.n_cols + j];
    if (h == h) { s1 += (double)h; s2 += (double)h * (double)h; ++cnt; }
  }
  for (int r = 0; r < A.n_rows; ++r) {
    const long long q = A.seq0 + r;
    float* cell = A.rows + (long long)r * A.ld + j;
    const float x = *cell;
    if (len == cap) {  // cannot happen with the trim below; kept for safety
      const float o = A.ring[((q - cap) % cap) * A.n_cols + j];
      if (o == o) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; }
      --len;
    }
    A.ring[(q % cap) * A.n_cols + j] = x;
    if (x == x) { s1 += (double)x; s2 += (double)x * (double)x; ++cnt; }
    ++len;
    if (q > 0) {  // the first row ever is returned as it came
      double out;
      if (cnt == 0) {
        out = NAN;
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
          out = ((double)x - mean) / sd;
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
      if (o == o) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; }
      --len;
    }
  }
}

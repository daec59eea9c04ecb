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

#define NMX_RAWNORM_MEAN 1
#define NMX_RAWNORM_ZSCORE 2

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
};

#ifdef NMX_HOST_EMU
NMX_DEV double nmx_wave_sum_d(double v) { return v; }
#else
NMX_DEV double nmx_wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif

// one wave per channel
NMX_DEV void nmx_rawnorm_stats_item(const NmxRawNormArgs& A, int c) {
  float* ring = A.ring + (long long)c * A.cap;
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
      a1 += (double)v; a2 += (double)v * (double)v;
    }
    s1 += nmx_wave_sum_d(a1); s2 += nmx_wave_sum_d(a2);
    cnt += n_new; len += n_new;
    NMX_SYNC();
    float mean = 0.f, scale = 0.f;
    if (!first) {
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
      s1 -= nmx_wave_sum_d(d1); s2 -= nmx_wave_sum_d(d2);
      len -= drop;
    }
  }
  if (NMX_TID == 0) { A.count[c] = cnt; A.len[c] = len; }
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
  if (sc != 0.f) {
    out = (x - A.mean[(long long)w * A.n_channels + c]) * sc;
    if (A.clip > 0.f) out = out < -A.clip ? -A.clip : (out > A.clip ? A.clip : out);
    out = nmx_clean(out);
  }
  A.y[idx] = out;
}

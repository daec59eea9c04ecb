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
// the scikit-learn based methods (processing/normalization.py:57-70,166-186: the scaler is FITTED on
// nan_to_num(history) every hop and transforms the current row).  Restated from scikit-learn's published
// algorithms (oracle/nm_oracle.py: _sk_fit_transform, pinned against the reference in tests/golden/norm_methods.npz):
#define NMX_NORM_ROBUST 4      // RobustScaler: (x - nanmedian) / (percentile 75 - 25), scale < 10 eps -> 1
#define NMX_NORM_MINMAX 5      // MinMaxScaler: x * s + (0 - min * s), s = 1 / (max - min), range < 10 eps -> 1
#define NMX_NORM_QUANTILE 6    // QuantileTransformer(n_quantiles = 300), uniform output

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
// scikit-learn methods: the history is nan_to_num'ed in FLOAT64 by the reference, so +-inf features count as
// +-DBL_MAX there; the fp32 sorted copy holds +-FLT_MAX for them -- widened back when read (WIDE)
NMX_DEV double nmx_norm_wide(float v) {
  return v >= 3.402823466e+38f ? 1.7976931348623157e308 : (v <= -3.402823466e+38f ? -1.7976931348623157e308 : (double)v);
}
template <bool WIDE = false>
NMX_DEV double nmx_norm_at(const float* S, int n_cols, int j, int k) {
  const float v = S[(long long)k * n_cols + j];
  return WIDE ? nmx_norm_wide(v) : (double)v;
}
template <bool WIDE = false>
NMX_DEV double nmx_norm_median(const float* S, int n_cols, int j, int n) {
  if (n == 0) return NAN;
  const double hi = nmx_norm_at<WIDE>(S, n_cols, j, n >> 1);
  return (n & 1) ? hi : 0.5 * (nmx_norm_at<WIDE>(S, n_cols, j, (n >> 1) - 1) + hi);
}

// np.percentile(S, 100 q) of the sorted column (method "linear", NumPy's _lerp: a + (b - a) t, from the upper end
// when t >= 0.5); n >= 1
NMX_DEV double nmx_norm_quantile(const float* S, int n_cols, int j, int n, double q) {
  const double vi = (double)(n - 1) * q;
  if (vi >= (double)(n - 1)) return nmx_norm_at<true>(S, n_cols, j, n - 1);
  if (vi < 0.0) return nmx_norm_at<true>(S, n_cols, j, 0);
  const double fl = floor(vi);
  const int lo = (int)fl;
  const double t = vi - fl;
  const double a = nmx_norm_at<true>(S, n_cols, j, lo), b = nmx_norm_at<true>(S, n_cols, j, lo + 1);
  const double d = b - a;
  return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}
// i-th of the nq quantiles QuantileTransformer fits: references = linspace(0, 1, nq), percentile(references * 100)
NMX_DEV double nmx_norm_ref(int i, int nq) { return (i < nq - 1) ? (double)i * (1.0 / (double)(nq - 1)) : 1.0; }
NMX_DEV double nmx_norm_qt(const float* S, int n_cols, int j, int n, int i, int nq) {
  if (nq == 1) return nmx_norm_at<true>(S, n_cols, j, 0);
  return nmx_norm_quantile(S, n_cols, j, n, nmx_norm_ref(i, nq) * 100.0 / 100.0);
}
NMX_DEV double nmx_norm_sklearn(const NmxNormArgs& A, int j, int n, double x) {
  const float* S = A.sorted;
  const int nc = A.n_cols;
  const double eps10 = 10.0 * 2.220446049250313e-16;
  if (A.method == NMX_NORM_ROBUST) {
    const double c = nmx_norm_median<true>(S, nc, j, n);
    double sc = nmx_norm_quantile(S, nc, j, n, 0.75) - nmx_norm_quantile(S, nc, j, n, 0.25);
    if (sc < eps10) sc = 1.0;
    return (x - c) / sc;
  }
  if (A.method == NMX_NORM_MINMAX) {
    const double lo = nmx_norm_at<true>(S, nc, j, 0), hi = nmx_norm_at<true>(S, nc, j, n - 1);
    double rng = hi - lo;
    if (rng < eps10) rng = 1.0;
    const double sc = 1.0 / rng;
    return x * sc + (0.0 - lo * sc);
  }
  // quantile: y = (interp(x, Q, R) - interp(-x, -Q[::-1], -R[::-1])) / 2; x == Q[0] -> 0, x == Q[-1] -> 1
  if (x != x) return x;
  const int nq = n < 300 ? n : 300;
  const double q0 = nmx_norm_qt(S, nc, j, n, 0, nq), q1 = nmx_norm_qt(S, nc, j, n, nq - 1, nq);
  if (x == q0) return 0.0;
  if (x == q1) return 1.0;
  if (x < q0) return 0.0;
  if (x > q1) return 1.0;
  // a: LAST index with Q[a] <= x (ascending interp);  b: FIRST index with Q[b] >= x (the reversed, negated one)
  int lo = 0, hi = nq - 1;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (nmx_norm_qt(S, nc, j, n, mid, nq) <= x) lo = mid; else hi = mid; }
  const int a = lo;
  lo = 0; hi = nq - 1;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (nmx_norm_qt(S, nc, j, n, mid, nq) >= x) hi = mid; else lo = mid; }
  const int b = hi;
  double r1, r2;
  {
    const double xa = nmx_norm_qt(S, nc, j, n, a, nq), ra = nmx_norm_ref(a, nq);
    if (xa == x) r1 = ra;
    else {
      const double xb = nmx_norm_qt(S, nc, j, n, a + 1, nq), rb = nmx_norm_ref(a + 1, nq);
      r1 = (rb - ra) / (xb - xa) * (x - xa) + ra;
    }
  }
  {   // xp = -Q[::-1], fp = -R[::-1], at -x: j' = nq - 1 - b
    const double xb = nmx_norm_qt(S, nc, j, n, b, nq), rb = nmx_norm_ref(b, nq);
    if (xb == x) r2 = -rb;
    else {
      const double xa = nmx_norm_qt(S, nc, j, n, b - 1, nq), ra = nmx_norm_ref(b - 1, nq);
      r2 = ((-ra) - (-rb)) / ((-xa) - (-xb)) * ((-x) - (-xb)) + (-rb);
    }
  }
  return 0.5 * (r1 - r2);
}

// 1 / a and 1 / sqrt(a) in float64 from the hardware seeds (v_rcp_f64 / v_rsq_f64) + two Newton steps: <= 1 ulp, a dozen
// instructions -- the IEEE division / square root sequences are ~30 quarter-rate instructions each, and the mean / z-score
// walk is issue bound on them (one wave per 64 columns: 156 waves on 1024 SIMDs).  The outputs are rounded to float32.
NMX_DEV double nmx_rcp_f64(double a) {
#ifdef NMX_HOST_EMU
  return 1.0 / a;
#else
  const double r0 = __builtin_amdgcn_rcp(a);
  if (!(fabs(r0) < INFINITY) || r0 == 0.0) return r0;   // a == 0, infinite or NaN: the seed is the answer
  double r = fma(fma(-a, r0, 1.0), r0, r0);
  r = fma(fma(-a, r, 1.0), r, r);
  return r;
#endif
}
NMX_DEV double nmx_rsq_f64(double a) {   // a > 0, finite
#ifdef NMX_HOST_EMU
  return 1.0 / sqrt(a);
#else
  double y = __builtin_amdgcn_rsq(a);
  y = y * fma(-0.5 * a * y, y, 1.5);
  y = y * fma(-0.5 * a * y, y, 1.5);
  return y;
#endif
}

#define NMX_NORM_PF 8
NMX_DEV void nmx_norm_column(const NmxNormArgs& A, int j) {
  if (j >= A.n_cols) return;
  if (A.colmask && !A.colmask[j]) return;
  const int cap = A.cap;
  // rebuild the sums from the history (the last min(seq0, cap - 1) rows)
  const long long have = A.seq0 < (long long)(cap - 1) ? A.seq0 : (long long)(cap - 1);
  double s1 = 0.0, s2 = 0.0;
  int cnt = 0, ninf = 0, len = (int)have;   // cnt: finite values in the window, ninf: +-inf values
  {   // (eight independent loads at a time, the slot advancing without a 64-bit modulo: the rebuild was 0.07 ms of a launch)
    int sl = (int)((A.seq0 - have) % cap);
    for (long long q0 = 0; q0 < have; q0 += 8) {
      float hb[8];
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int i = 0; i < 8; ++i) {
        hb[i] = q0 + i < have ? A.ring[(long long)sl * A.n_cols + j] : NAN;
        sl = sl + 1 == cap ? 0 : sl + 1;
      }
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int i = 0; i < 8; ++i) {
        const float h = hb[i];
        if (q0 + i >= have) continue;
        if (nmx_norm_finite(h)) { s1 += (double)h; s2 += (double)h * (double)h; ++cnt; }
        else if (h == h) ++ninf;
      }
    }
  }
  const bool med = A.method >= NMX_NORM_MEDIAN;
  const bool sk = A.method >= NMX_NORM_ROBUST;   // history = nan_to_num(history): EVERY row is in the sorted copy
  int ns = 0;   // entries of the sorted copy (== cnt; sk: == len)
  if (med)
    for (long long q = A.seq0 - have; q < A.seq0; ++q) {
      const float h = A.ring[(q % cap) * A.n_cols + j];
      if (sk) nmx_norm_insert(A.sorted, A.n_cols, j, ns, nmx_clean(h));
      else if (h == h) nmx_norm_insert(A.sorted, A.n_cols, j, ns, h);
    }
  bool pend = false;   // median methods: the value trimmed after the previous hop leaves the sorted copy
  float pend_val = 0.f;   //   together with the next insertion (nothing reads the copy in between)
  // Rows in blocks of NMX_NORM_PF: the block's cells and the values that will leave the history during it are loaded
  // up front (independent loads).  One thread walks a column hop by hop, and with the loads inside the walk every hop
  // paid two dependent global-memory round trips -- 3 us per row, 0.8 ms per 256-hop chunk of the default stream for
  // 124 waves of work.  (The slot a row trims was written cap - 1 rows earlier: before the block when cap - 1 >= PF.)
  const bool pf_o = cap - 1 >= NMX_NORM_PF;
  if (!med) {
    // "mean" / "zscore" (the default): the walk splits into what IS sequential -- the sliding sums, a handful of float64
    // additions per hop -- and what is not: a hop's mean, standard deviation and quotient (float64 divisions and a square
    // root: ~150 dependent instructions) only read that hop's sums.  Per block of NMX_NORM_PF hops the sums are advanced
    // first (snapshots in registers), then the block's outputs are formed side by side: one thread per column keeps eight
    // division chains in flight instead of one (the kernel is 156 waves of pure latency; measured in profiles/README.md, round 5).
    // (the walk itself holds no division and no 64-bit modulo: the ring slot advances with the hops -- the value a hop trims,
    // row q - (cap - 1), sits in the slot the NEXT hop writes --, and the trim's tests are multiplied through by the count)
    // (Loading the NEXT block's cells while this block is computed was measured: nothing standalone -- the walk is a chain
    // of dependent float64 instructions, ~1600 cycles per hop, not load latency -- and worse inside the plan, 8.02 -> 8.40 ms
    // per 1024 hops with the z-score: the loads in flight compete with the next chunk's kernels.)
    int slot = (int)(A.seq0 % cap);
    float xb[NMX_NORM_PF], ob[NMX_NORM_PF];
    auto load_block = [&](int rb, int slot0, float (&xv)[NMX_NORM_PF], float (&ov)[NMX_NORM_PF]) {
      int so = slot0;
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int i = 0; i < NMX_NORM_PF; ++i) {
        const int rr = rb + i < A.n_rows ? rb + i : A.n_rows - 1;
        xv[i] = A.rows[(long long)rr * A.ld + j];
        so = so + 1 == cap ? 0 : so + 1;   // slot of hop rb + i + 1 == slot of row (rb + i) - (cap - 1)
        const long long qo = A.seq0 + rr - (cap - 1);
        ov[i] = (pf_o && qo >= 0) ? A.ring[(long long)so * A.n_cols + j] : 0.f;
      }
    };
    load_block(0, slot, xb, ob);
    for (int r0 = 0; r0 < A.n_rows; r0 += NMX_NORM_PF) {
      double S1[NMX_NORM_PF], S2[NMX_NORM_PF], VO[NMX_NORM_PF];   // sums after the hop's value entered; two-pass variance or < 0
      int CN[NMX_NORM_PF], NI[NMX_NORM_PF];
      const bool more = r0 + NMX_NORM_PF < A.n_rows;
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int bi = 0; bi < NMX_NORM_PF; ++bi) {
        S1[bi] = 0.0; S2[bi] = 0.0; VO[bi] = -1.0; CN[bi] = 0; NI[bi] = 0;
        if (r0 + bi >= A.n_rows) continue;
        const long long q = A.seq0 + r0 + bi;
        const float x = xb[bi];
        const int next = slot + 1 == cap ? 0 : slot + 1;
        if (len == cap) {  // cannot happen with the trim below; kept for safety
          const float o = A.ring[(long long)slot * A.n_cols + j];
          if (o == o) {
            if (nmx_norm_finite(o)) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; } else --ninf;
          }
          --len;
        }
        A.ring[(long long)slot * A.n_cols + j] = x;
        {   // (selects, not branches: the walk is a chain of a few dozen dependent instructions per hop, a skipped branch
            //  costs as much as what it skips)
          const bool fx = nmx_norm_finite(x);
          const double xd = fx ? (double)x : 0.0;
          s1 += xd; s2 += xd * xd;
          cnt += fx ? 1 : 0;
          ninf += (!fx && x == x) ? 1 : 0;
        }
        ++len;
        S1[bi] = s1; S2[bi] = s2; CN[bi] = cnt; NI[bi] = ninf;
        // cancellation (one-pass error ~ eps mean^2 / var; var < 1e-8 mean^2 without a division: x cnt^2): two-pass
        // over the ring NOW, while it holds this hop's window (rare)
        if (A.method == NMX_NORM_ZSCORE && q > 0 && cnt > 0 && ninf == 0 && s2 * (double)cnt - s1 * s1 < 1e-8 * s1 * s1) {
          const double mean = s1 / (double)cnt;
          double acc = 0.0;
          for (long long t = q - len + 1; t <= q; ++t) {
            const float h = A.ring[(t % cap) * A.n_cols + j];
            if (h == h) { const double d = (double)h - mean; acc += d * d; }
          }
          VO[bi] = acc / (double)cnt;
        }
        // history keeps its last N - 1 rows (normalization.py:107)
        if (len > cap - 1) {
          const float o = pf_o ? ob[bi] : A.ring[(long long)next * A.n_cols + j];
          {
            const bool fo = nmx_norm_finite(o);
            const double od = fo ? (double)o : 0.0;
            s1 -= od; s2 -= od * od;
            cnt -= fo ? 1 : 0;
            ninf -= (!fo && o == o) ? 1 : 0;
          }
          --len;
          // (a value far larger than what stays behind leaves the sums with ITS rounding: rebuild them -- see the general
          // walk below for the measure: o^2 <= 1e4 s2 and |o| (|o| + 2 |mean|) <= 1e5 (s2 - s1^2 / cnt), times cnt)
          if (nmx_norm_finite(o)) {
            const double oabs = fabs((double)o), c = (double)cnt;
            const bool keep = cnt > 0 && oabs * oabs <= 1e4 * s2 && oabs * (oabs * c + 2.0 * fabs(s1)) <= 1e5 * (s2 * c - s1 * s1);
            if (!keep) {
              s1 = 0.0; s2 = 0.0; cnt = 0; ninf = 0;
              for (long long t = q - len + 1; t <= q; ++t) {
                const float h = A.ring[(t % cap) * A.n_cols + j];
                if (nmx_norm_finite(h)) { s1 += (double)h; s2 += (double)h * (double)h; ++cnt; }
                else if (h == h) ++ninf;
              }
            }
          }
        }
        slot = next;
      }
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
      for (int bi = 0; bi < NMX_NORM_PF; ++bi) {   // the block's outputs: independent of each other
        if (r0 + bi >= A.n_rows) continue;
        const long long q = A.seq0 + r0 + bi;
        if (q == 0) continue;   // the first row ever is returned as it came
        const double x = (double)xb[bi];
        double out;
        if (CN[bi] + NI[bi] == 0 || NI[bi] > 0) {
          out = NAN;   // empty window, or +-inf inside it: mean +-inf / NaN, std NaN (see the header)
        } else {
          // sums / count as CORRECTLY rounded quotients (seed product + one residual step, the tail of the IEEE division
          // sequence): a constant column must give mean == x and variance == 0 exactly, as numpy.mean / numpy.std do
          const double c = (double)CN[bi], rc = nmx_rcp_f64(c);
          double mean = S1[bi] * rc;
          mean = fma(fma(-mean, c, S1[bi]), rc, mean);
          if (A.method == NMX_NORM_MEAN) {
            out = (x - mean) * nmx_rcp_f64(mean);
          } else {
            double ex2 = S2[bi] * rc;
            ex2 = fma(fma(-ex2, c, S2[bi]), rc, ex2);
            const double var = VO[bi] >= 0.0 ? VO[bi] : ex2 - mean * mean;
            out = (x - mean) * (var > 0.0 ? nmx_rsq_f64(var) : 1.0);   // (std 0 -> 1, normalization.py:160-164)
          }
        }
        if (A.clip > 0.f) {  // ndarray.clip: NaN stays NaN
          if (out < -(double)A.clip) out = -(double)A.clip;
          if (out > (double)A.clip) out = (double)A.clip;
        }
        A.rows[(long long)(r0 + bi) * A.ld + j] = nmx_clean((float)out);
      }
      if (more) load_block(r0 + NMX_NORM_PF, slot, xb, ob);
    }
    return;
  }
  for (int r0 = 0; r0 < A.n_rows; r0 += NMX_NORM_PF) {
  float xb[NMX_NORM_PF], ob[NMX_NORM_PF];
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
  for (int i = 0; i < NMX_NORM_PF; ++i) {
    const int rr = r0 + i < A.n_rows ? r0 + i : A.n_rows - 1;
    xb[i] = A.rows[(long long)rr * A.ld + j];
    const long long qo = A.seq0 + rr - (cap - 1);
    ob[i] = (pf_o && qo >= 0) ? A.ring[(qo % cap) * A.n_cols + j] : 0.f;
  }
  for (int bi = 0; bi < NMX_NORM_PF && r0 + bi < A.n_rows; ++bi) {
    const int r = r0 + bi;
    const long long q = A.seq0 + r;
    float* cell = A.rows + (long long)r * A.ld + j;
    const float x = xb[bi];
    if (len == cap) {  // cannot happen with the trim below; kept for safety
      const float o = A.ring[((q - cap) % cap) * A.n_cols + j];
      if (o == o) {
        if (nmx_norm_finite(o)) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; } else --ninf;
      }
      if (med && (sk || o == o)) {
        if (pend) { nmx_norm_remove(A.sorted, A.n_cols, j, ns, pend_val); pend = false; }
        nmx_norm_remove(A.sorted, A.n_cols, j, ns, sk ? nmx_clean(o) : o);
      }
      --len;
    }
    A.ring[(q % cap) * A.n_cols + j] = x;
    if (nmx_norm_finite(x)) { s1 += (double)x; s2 += (double)x * (double)x; ++cnt; }
    else if (x == x) ++ninf;
    if (med) {
      const bool in_s = sk || x == x;
      const float xs = sk ? nmx_clean(x) : x;
      if (in_s && pend) nmx_norm_replace(A.sorted, A.n_cols, j, ns, pend_val, xs);
      else if (in_s) nmx_norm_insert(A.sorted, A.n_cols, j, ns, xs);
      else if (pend) nmx_norm_remove(A.sorted, A.n_cols, j, ns, pend_val);
      pend = false;
    }
    ++len;
    if (q > 0) {  // the first row ever is returned as it came
      double out;
      if (sk) {
        // (the current value is widened like the history: a nan_to_num'ed -inf feature is -DBL_MAX on both sides of
        // the reference's subtraction, not -FLT_MAX against -DBL_MAX)
        out = nmx_norm_sklearn(A, j, ns, nmx_norm_finite(x) ? nmx_norm_wide(x) : (double)x);
      } else if (cnt + ninf == 0 || (ninf > 0 && A.method != NMX_NORM_MEDIAN)) {
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
          if (var < 1e-8 * mean * mean) {  // cancellation (one-pass error ~ eps mean^2 / var): two-pass over the ring (rare)
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
      const float o = pf_o ? ob[bi] : A.ring[((q - (cap - 1)) % cap) * A.n_cols + j];
      if (o == o) {
        if (nmx_norm_finite(o)) { s1 -= (double)o; s2 -= (double)o * (double)o; --cnt; } else --ninf;
      }
      if (med && (sk || o == o)) { pend = true; pend_val = sk ? nmx_clean(o) : o; }
      --len;
      // a value far larger than what stays behind leaves the sliding sums with ITS rounding (4e1 leaving a window of
      // 1e-5's: s2 is 1e-13 off against 7e-9): rebuild the sums from the rows that remain (rare).  "Far larger" is
      // measured against the SPREAD that stays (cnt x variance = s2 - s1^2 / cnt), not against s2: -13.1 leaving
      // four values of -0.2023 +- 5e-5 left 2e-14 in s2 against 8e-9 -- the z-scores of the next hops were 7e-6 off
      // (and differed by as much between a batch call and window-by-window calls, which rebuild: fuzz seed 5195)
      // (its residue in cnt x variance = s2 - s1^2 / cnt is ~ eps (o^2 + 2 |mean| |o|) per addition it sat through)
      const double spread = cnt > 0 ? s2 - s1 * s1 / (double)cnt : 0.0;
      const double mabs = cnt > 0 ? fabs(s1) / (double)cnt : 0.0, oabs = fabs((double)o);
      if (nmx_norm_finite(o) && (!(oabs * oabs <= 1e4 * s2) || !(oabs * (oabs + 2.0 * mabs) <= 1e5 * spread))) {
        s1 = 0.0; s2 = 0.0; cnt = 0; ninf = 0;
        for (long long t = q - len + 1; t <= q; ++t) {
          const float h = A.ring[(t % cap) * A.n_cols + j];
          if (nmx_norm_finite(h)) { s1 += (double)h; s2 += (double)h * (double)h; ++cnt; }
          else if (h == h) ++ninf;
        }
      }
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// "mean" / "zscore" over a batch as SCANS (round 6).  The column walk above is one dependent chain per column: sliding
// sums, then ~150 float64 instructions of division / square root per hop -- 156 waves of pure latency for the default
// stream (9984 columns), 0.87 ms per 1024 hops inside the plan.  Nothing but the sums is sequential, and with at most
// cap - 1 hops per launch the window of hop r is a SUFFIX of the history the launch found plus a PREFIX of the batch:
//     S(r) = Bk[d(r)] + F[r],   Bk[d] = sum of history rows d .. have - 1,   F[r] = sum of batch rows 0 .. r,
//                               d(r) = max(0, r - (cap - 1 - have))
// (the van Herk / Gil-Werman decomposition with the block boundary at the start of the batch).  Additions only: a value that
// leaves the window is simply not summed, so the walk's "a value far larger than what stays behind leaves its rounding in
// the sums: rebuild" logic has nothing to do here.  Both scans are cut into segments of NMX_NORM_SEG rows -- one THREAD
// per (column, segment): 32 independent loads, 32 additions --, a third kernel turns the segments' totals into offsets:
//   nmx_norm_seg_hist    local suffix sums of a history segment -> the scratch row of the hop they belong to; total -> table
//   nmx_norm_seg_batch   local prefix sums of a batch segment + the hop's history part -> scratch; raw values -> scratch
//   nmx_norm_seg_offsets per column: exclusive suffix (history) / prefix (batch) sums of the tables, in place (<= 20 steps)
//   nmx_norm_scan_cell   one thread per (hop, column): sums = local + offsets, then mean, standard deviation, quotient,
//                        clip, nan_to_num -- the expensive part, 2.5 M independent cells per 256 hops, not 9984 chains
//   nmx_norm_scan_ring   the batch's raw values into the ring (AFTER the cells: a hop's two-pass variance reads history
//                        rows the batch overwrites).
// Same results as the walk up to the association of float64 sums of <= cap float32 values (1e-16 relative).
// (The first form of this had ONE thread scan a column's 555 rows: 0.10 ms alone, 0.29 - 0.40 ms next to the following
// chunk's kernels -- a long dependent chain is what co-running waves slow down most.)
#define NMX_NORM_SEG 32
struct NmxNormScan {
  double* s1;        // [n_rows + 1][n_cols] sums of the finite values: local parts (row n_rows: history position 0)
  double* s2;        // ... of their squares
  unsigned* cn;      // finite count | (+-inf count << 16)
  float* xs;         // [n_rows][n_cols] the batch's raw values
  double* t1;        // [n_hseg + n_bseg][n_cols] segment totals, then offsets (history segments first)
  double* t2;
  unsigned* tc;
  int have, grow, dmax, n_hseg, n_bseg;
};

NMX_DEV bool nmx_norm_scan_ok(const NmxNormArgs& A) {
  return (A.method == NMX_NORM_MEAN || A.method == NMX_NORM_ZSCORE) && A.n_rows >= 1 && A.n_rows <= A.cap - 1 && A.cap < 65536;
}
// (host) the launch geometry of one piece
static inline void nmx_norm_scan_shape(const NmxNormArgs& A, NmxNormScan& Sc) {
  Sc.have = (int)(A.seq0 < (long long)(A.cap - 1) ? A.seq0 : (long long)(A.cap - 1));
  Sc.grow = A.cap - 1 - Sc.have;                                            // hops before the history starts to lose rows
  Sc.dmax = A.n_rows - 1 - Sc.grow > 0 ? A.n_rows - 1 - Sc.grow : 0;        // d(n_rows - 1) <= have - 1
  Sc.n_hseg = (Sc.have + NMX_NORM_SEG - 1) / NMX_NORM_SEG;
  Sc.n_bseg = (A.n_rows + NMX_NORM_SEG - 1) / NMX_NORM_SEG;
}

// history positions [g SEG, (g + 1) SEG) of column j, newest -> oldest (position i = row seq0 - have + i)
NMX_DEV void nmx_norm_seg_hist(const NmxNormArgs& A, const NmxNormScan& Sc, int g, int j) {
  if (j >= A.n_cols || g >= Sc.n_hseg) return;
  if (A.colmask && !A.colmask[j]) return;
  const int cap = A.cap, nc = A.n_cols;
  const int lo = g * NMX_NORM_SEG, hi = lo + NMX_NORM_SEG < Sc.have ? lo + NMX_NORM_SEG : Sc.have;
  float hb[NMX_NORM_SEG];
  int sl = (int)((A.seq0 - Sc.have + lo) % cap);
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
  for (int k = 0; k < NMX_NORM_SEG; ++k) {
    hb[k] = lo + k < hi ? A.ring[(long long)sl * nc + j] : NAN;
    sl = sl + 1 == cap ? 0 : sl + 1;
  }
  double b1 = 0.0, b2 = 0.0;
  int bc = 0, bi = 0;
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
  for (int k = NMX_NORM_SEG - 1; k >= 0; --k) {
    const int i = lo + k;
    if (i >= hi) continue;
    const float h = hb[k];
    const bool f = nmx_norm_finite(h);
    const double hd = f ? (double)h : 0.0;
    b1 += hd; b2 += hd * hd;
    bc += f ? 1 : 0;
    bi += (!f && h == h) ? 1 : 0;
    if (i <= Sc.dmax) {   // position i >= 1 belongs to hop i + grow, position 0 to hops 0 .. grow (row n_rows of the scratch)
      const long long o = (long long)(i >= 1 ? i + Sc.grow : A.n_rows) * nc + j;
      Sc.s1[o] = b1; Sc.s2[o] = b2; Sc.cn[o] = (unsigned)bc | ((unsigned)bi << 16);
    }
  }
  const long long t = (long long)g * nc + j;
  Sc.t1[t] = b1; Sc.t2[t] = b2; Sc.tc[t] = (unsigned)bc | ((unsigned)bi << 16);
}

// batch rows [g SEG, (g + 1) SEG) of column j, oldest -> newest
NMX_DEV void nmx_norm_seg_batch(const NmxNormArgs& A, const NmxNormScan& Sc, int g, int j) {
  if (j >= A.n_cols || g >= Sc.n_bseg) return;
  if (A.colmask && !A.colmask[j]) return;
  const int nc = A.n_cols, n = A.n_rows;
  const int lo = g * NMX_NORM_SEG, hi = lo + NMX_NORM_SEG < n ? lo + NMX_NORM_SEG : n;
  float xb[NMX_NORM_SEG];
  double p1[NMX_NORM_SEG], p2[NMX_NORM_SEG];
  unsigned pc[NMX_NORM_SEG];
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
  for (int k = 0; k < NMX_NORM_SEG; ++k) {
    const int r = lo + k < hi ? lo + k : hi - 1;
    xb[k] = A.rows[(long long)r * A.ld + j];
    const long long o = (long long)(r > Sc.grow ? r : n) * nc + j;   // the hop's history part (local to its segment)
    p1[k] = Sc.have ? Sc.s1[o] : 0.0; p2[k] = Sc.have ? Sc.s2[o] : 0.0; pc[k] = Sc.have ? Sc.cn[o] : 0u;
  }
  double f1 = 0.0, f2 = 0.0;
  int fc = 0, fi = 0;
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
  for (int k = 0; k < NMX_NORM_SEG; ++k) {
    const int r = lo + k;
    if (r >= hi) continue;
    const float x = xb[k];
    const bool f = nmx_norm_finite(x);
    const double xd = f ? (double)x : 0.0;
    f1 += xd; f2 += xd * xd;
    fc += f ? 1 : 0;
    fi += (!f && x == x) ? 1 : 0;
    const long long o = (long long)r * nc + j;
    Sc.s1[o] = p1[k] + f1; Sc.s2[o] = p2[k] + f2;
    Sc.cn[o] = ((pc[k] & 0xffffu) + (unsigned)fc) | (((pc[k] >> 16) + (unsigned)fi) << 16);
    Sc.xs[o] = x;
  }
  const long long t = (long long)(Sc.n_hseg + g) * nc + j;
  Sc.t1[t] = f1; Sc.t2[t] = f2; Sc.tc[t] = (unsigned)fc | ((unsigned)fi << 16);
}

// totals -> offsets: history segment g gets the sum of the segments NEWER than it (g' > g), batch segment g of the older ones
NMX_DEV void nmx_norm_seg_offsets(const NmxNormArgs& A, const NmxNormScan& Sc, int j) {
  if (j >= A.n_cols) return;
  if (A.colmask && !A.colmask[j]) return;
  const int nc = A.n_cols;
  double a1 = 0.0, a2 = 0.0;
  unsigned ac = 0u;
  for (int g = Sc.n_hseg - 1; g >= 0; --g) {
    const long long t = (long long)g * nc + j;
    const double v1 = Sc.t1[t], v2 = Sc.t2[t];
    const unsigned vc = Sc.tc[t];
    Sc.t1[t] = a1; Sc.t2[t] = a2; Sc.tc[t] = ac;
    a1 += v1; a2 += v2; ac += vc;   // (16-bit fields: counts <= cap < 65536 never carry)
  }
  a1 = 0.0; a2 = 0.0; ac = 0u;
  for (int g = 0; g < Sc.n_bseg; ++g) {
    const long long t = (long long)(Sc.n_hseg + g) * nc + j;
    const double v1 = Sc.t1[t], v2 = Sc.t2[t];
    const unsigned vc = Sc.tc[t];
    Sc.t1[t] = a1; Sc.t2[t] = a2; Sc.tc[t] = ac;
    a1 += v1; a2 += v2; ac += vc;
  }
}

NMX_DEV void nmx_norm_scan_cell(const NmxNormArgs& A, const NmxNormScan& Sc, int r, int j) {
  if (j >= A.n_cols || r >= A.n_rows) return;
  if (A.colmask && !A.colmask[j]) return;
  const long long q = A.seq0 + r;
  if (q == 0) return;   // the first row ever is returned as it came
  const int nc = A.n_cols;
  const long long o = (long long)r * nc + j;
  const double x = (double)Sc.xs[o];
  const long long tb = (long long)(Sc.n_hseg + r / NMX_NORM_SEG) * nc + j;
  double S1 = Sc.t1[tb], S2 = Sc.t2[tb];
  unsigned cn = Sc.tc[tb];
  if (Sc.have) {
    const int d = r > Sc.grow ? r - Sc.grow : 0;
    const long long th = (long long)(d / NMX_NORM_SEG) * nc + j;
    S1 += Sc.t1[th]; S2 += Sc.t2[th]; cn += Sc.tc[th];
  }
  S1 += Sc.s1[o]; S2 += Sc.s2[o]; cn += Sc.cn[o];
  const int cnt = (int)(cn & 0xffffu), ninf = (int)(cn >> 16);
  double out;
  if (cnt + ninf == 0 || ninf > 0) {
    out = NAN;   // empty window, or +-inf inside it: mean +-inf / NaN, std NaN (see the header)
  } else {
    // sums / count as CORRECTLY rounded quotients (nmx_norm_column): a constant column gives mean == x, variance == 0
    const double c = (double)cnt, rc = nmx_rcp_f64(c);
    double mean = S1 * rc;
    mean = fma(fma(-mean, c, S1), rc, mean);
    if (A.method == NMX_NORM_MEAN) {
      out = (x - mean) * nmx_rcp_f64(mean);
    } else {
      double ex2 = S2 * rc;
      ex2 = fma(fma(-ex2, c, S2), rc, ex2);
      double var = ex2 - mean * mean;
      if (S2 * c - S1 * S1 < 1e-8 * S1 * S1) {
        // cancellation (one-pass error ~ eps mean^2 / var): two passes over the window -- history rows from the ring
        // (untouched until nmx_norm_scan_ring), the batch's rows from the scratch copy (rare)
        const int cap = A.cap;
        long long t0 = q - (cap - 1);
        if (t0 < A.seq0 - Sc.have) t0 = A.seq0 - Sc.have;
        double acc = 0.0;
        for (long long t = t0; t <= q; ++t) {
          const float h = t < A.seq0 ? A.ring[(t % cap) * nc + j] : Sc.xs[(t - A.seq0) * nc + j];
          if (h == h) { const double d = (double)h - mean; acc += d * d; }
        }
        var = acc / c;
      }
      out = (x - mean) * (var > 0.0 ? nmx_rsq_f64(var) : 1.0);   // (std 0 -> 1, normalization.py:160-164)
    }
  }
  if (A.clip > 0.f) {  // ndarray.clip: NaN stays NaN
    if (out < -(double)A.clip) out = -(double)A.clip;
    if (out > (double)A.clip) out = (double)A.clip;
  }
  A.rows[(long long)r * A.ld + j] = nmx_clean((float)out);
}

NMX_DEV void nmx_norm_scan_ring(const NmxNormArgs& A, const NmxNormScan& Sc, int r, int j) {
  if (j >= A.n_cols || r >= A.n_rows) return;
  if (A.colmask && !A.colmask[j]) return;
  A.ring[((A.seq0 + r) % A.cap) * A.n_cols + j] = Sc.xs[(long long)r * A.n_cols + j];
}

// nmx_k_power.h -- the "power" normaliser: scikit-learn's PowerTransformer (Yeo-Johnson, standardize=True) FITTED on
// nan_to_num(history) every hop and applied to the current row (processing/normalization.py:57-70,172-186).
//
// What the reference computes per feature column and hop (scikit-learn 1.7 / scipy >= 1.9, restated in
// oracle/nm_oracle.py: yeo_johnson_fit, checked there against PowerTransformer itself):
//   lambda = scipy.stats.yeojohnson_normmax(column): a bounded Brent search (scipy.optimize.fminbound, xtol 1.48e-8,
//            <= 500 evaluations) of the log-likelihood  -n/2 log var(T_lambda(x)) + (lambda - 1) sum sign(x) log1p|x|,
//            T in scipy's expm1 / log1p form, bounds from log1p(20 max|x|); lambda = 1 for a constant column;
//   out    = (T'_lambda(x_now) - mean(T'_lambda(history))) / std(T'_lambda(history)),  T' = scikit-learn's np.power form.
// ~30 likelihood evaluations x the history length (<= 300 rows at the defaults) transcendental operations per VALUE:
// the reference spends about a second per hop on a 10 000-column row.  Here every (hop, column) pair of a batch is
// ONE THREAD -- the history of a hop is known from the batch itself, so the hops are independent -- in float64
// (the likelihood is flat around its maximum: an fp32 evaluation would leave lambda at 1e-3), with
// sign(x) log1p|x| of every history value computed once per batch (it does not depend on lambda) instead of once per
// evaluation.  Lanes = consecutive columns: every history access is coalesced.
#pragma once

#include "nmx_device.h"

#define NMX_NORM_POWER 7

struct NmxPowerArgs {
  // extended history of the batch: row e < have = the carried history (oldest first), row have + r = batch row r
  const float* ext;            // [have + n_rows][n_cols] raw fp32 values
  const double* sl;            // [have + n_rows][n_cols] sign(x) log1p|x| of the nan_to_num'ed values
  float* rows;                 // [n_rows][ld] in place (read from ext, written here)
  long long ld;
  int n_rows, n_cols, have;
  int cap;                     // N: rows of the window incl. the current one
  long long seq0;              // rows seen before this batch
  const unsigned char* colmask;
  float clip;
};

NMX_DEV double nmx_pw_clean(float v) {   // np.nan_to_num in float64
  if (v != v) return 0.0;
  if (v > 3.402823466e+38f) return 1.7976931348623157e308;
  if (v < -3.402823466e+38f) return -1.7976931348623157e308;
  return (double)v;
}
NMX_DEV double nmx_pw_sl(float v) {
  const double x = nmx_pw_clean(v);
  const double l = log1p(fabs(x));
  return x > 0.0 ? l : (x < 0.0 ? -l : 0.0);   // np.sign(x) * log1p(|x|)
}

// one column of a window: n values at p[i * stride]
struct NmxPwCol {
  const float* x;
  const double* sl;
  long long stride;
  int n;
};

// -yeojohnson_llf(lmb) (scipy/stats/_morestats.py), +inf where the transformed variance underflows
NMX_DEV double nmx_pw_negllf(double lmb, const NmxPwCol& c, double sl_sum) {
  const double eps = 2.220446049250313e-16, tiny = 2.2250738585072014e-308;
  const bool l0 = fabs(lmb) < eps, l2 = !(fabs(lmb - 2.0) > eps);
  const double c2 = 2.0 - lmb;
  double k = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < c.n; ++i) {
    const double s = c.sl[(long long)i * c.stride], l = fabs(s);
    double t;
    if (s >= 0.0) t = l0 ? l : expm1(lmb * l) / lmb;
    else t = l2 ? -l : -expm1(c2 * l) / c2;
    if (i == 0) k = t;
    const double d = t - k;   // shifted one-pass variance (the reference: two passes; same to ~1e-15 relative)
    s1 += d;
    s2 += d * d;
  }
  double var = (s2 - s1 * s1 / (double)c.n) / (double)c.n;
  if (var < 0.0) var = 0.0;
  if (var < tiny) return INFINITY;
  const double llf = -(double)c.n / 2.0 * log(var) + (lmb - 1.0) * sl_sum;
  if (llf == INFINITY || llf == -INFINITY) return INFINITY;
  return -llf;
}

NMX_DEV double nmx_pw_sign1(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 1.0); }   // np.sign(v) + (v == 0)

// scipy.optimize._optimize._minimize_scalar_bounded (fminbound), statement for statement; f(lambda) = -log-likelihood
template <class F>
NMX_DEV double nmx_pw_fminbound_f(F f, double x1, double x2) {
  const double xatol = 1.48e-8, sqrt_eps = sqrt(2.2e-16), golden_mean = 0.5 * (3.0 - sqrt(5.0));
  double a = x1, b = x2;
  double fulc = a + golden_mean * (b - a);
  double nfc = fulc, xf = fulc;
  double rat = 0.0, e = 0.0;
  double x = xf;
  double fx = f(x);
  int num = 1;
  double ffulc = fx, fnfc = fx;
  double xm = 0.5 * (a + b);
  double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0, tol2 = 2.0 * tol1;
  while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
    bool golden = true;
    if (fabs(e) > tol1) {
      golden = false;
      double r = (xf - nfc) * (fx - ffulc);
      double q = (xf - fulc) * (fx - fnfc);
      double p = (xf - fulc) * q - (xf - nfc) * r;
      q = 2.0 * (q - r);
      if (q > 0.0) p = -p;
      q = fabs(q);
      r = e;
      e = rat;
      if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
        rat = (p + 0.0) / q;
        x = xf + rat;
        if (((x - a) < tol2) || ((b - x) < tol2)) rat = tol1 * nmx_pw_sign1(xm - xf);
      } else {
        golden = true;
      }
    }
    if (golden) {
      e = xf >= xm ? a - xf : b - xf;
      rat = golden_mean * e;
    }
    const double si = nmx_pw_sign1(rat);
    x = xf + si * (fabs(rat) > tol1 ? fabs(rat) : tol1);
    const double fu = f(x);
    ++num;
    if (fu <= fx) {
      if (x >= xf) a = xf; else b = xf;
      fulc = nfc; ffulc = fnfc;
      nfc = xf; fnfc = fx;
      xf = x; fx = fu;
    } else {
      if (x < xf) a = x; else b = x;
      if ((fu <= fnfc) || (nfc == xf)) {
        fulc = nfc; ffulc = fnfc;
        nfc = x; fnfc = fu;
      } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
        fulc = x; ffulc = fu;
      }
    }
    xm = 0.5 * (a + b);
    tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    tol2 = 2.0 * tol1;
    if (num >= 500) break;
  }
  return xf;
}

NMX_DEV double nmx_pw_fminbound(const NmxPwCol& c, double sl_sum, double x1, double x2) {
  return nmx_pw_fminbound_f([&](double l) { return nmx_pw_negllf(l, c, sl_sum); }, x1, x2);
}

// sklearn.preprocessing.PowerTransformer._yeo_johnson_transform (the np.power form) of one value
NMX_DEV double nmx_pw_transform(double x, double lmb) {
  const double eps = 2.220446049250313e-16;
  if (x >= 0.0) return fabs(lmb) < eps ? log1p(x) : (pow(x + 1.0, lmb) - 1.0) / lmb;
  if (fabs(lmb - 2.0) > eps) return -(pow(-x + 1.0, 2.0 - lmb) - 1.0) / (2.0 - lmb);
  return -log1p(-x);
}

NMX_DEV bool nmx_pw_constant(double var, double mean, int n) {   // sklearn _is_constant_feature
  const double eps = 2.220446049250313e-16, t = (double)n * mean * eps;
  return var <= (double)n * eps * var + t * t;
}

// (lambda, mean, scale) of a history column, as PowerTransformer(standardize=True).fit leaves them
NMX_DEV void nmx_pw_fit(const NmxPwCol& c, double& lmb, double& mean_t, double& scale_t) {
  const int n = c.n;
  // constant column? (np.mean / np.var of the raw values, float64)
  double s = 0.0, amax = 0.0, sl_sum = 0.0;
  bool any_neg = false, all_neg = true, all_zero = true;
  for (int i = 0; i < n; ++i) {
    const double x = nmx_pw_clean(c.x[(long long)i * c.stride]);
    s += x;
    const double ax = fabs(x);
    if (ax > amax) amax = ax;
    any_neg |= x < 0.0;
    all_neg &= x < 0.0;
    all_zero &= x == 0.0;
    sl_sum += c.sl[(long long)i * c.stride];
  }
  const double mean = s / (double)n;
  double q = 0.0;
  for (int i = 0; i < n; ++i) {
    const double d = nmx_pw_clean(c.x[(long long)i * c.stride]) - mean;
    q += d * d;
  }
  if (nmx_pw_constant(q / (double)n, mean, n) || all_zero) {
    lmb = 1.0;
  } else {
    // scipy.stats.yeojohnson_normmax, brack=None: the search interval from the largest |x|
    const double log_eps = log(2.220446049250313e-16);
    const double log1p_max_x = log1p(20.0 * amax);
    double lb = (log(2.2250738585072014e-308) - log_eps) / 2.0 / log1p_max_x;
    double ub = (log(1.7976931348623157e308) + log_eps) / 2.0 / log1p_max_x;
    if (all_neg) { const double t = lb; lb = 2.0 - ub; ub = 2.0 - t; }
    else if (any_neg) { const double l2 = 2.0 - ub, u2 = 2.0 - lb; lb = l2 > lb ? l2 : lb; ub = u2 < ub ? u2 : ub; }
    lmb = nmx_pw_fminbound(c, sl_sum, lb, ub);
  }
  // StandardScaler on the transformed history
  double k = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < n; ++i) {
    const double t = nmx_pw_transform(nmx_pw_clean(c.x[(long long)i * c.stride]), lmb);
    if (i == 0) k = t;
    const double d = t - k;
    s1 += d;
    s2 += d * d;
  }
  mean_t = k + s1 / (double)n;
  double var_t = (s2 - s1 * s1 / (double)n) / (double)n;
  if (var_t < 0.0) var_t = 0.0;
  scale_t = nmx_pw_constant(var_t, mean_t, n) ? 1.0 : sqrt(var_t);
}

// ---- kernels: one thread per (extended row, column) / (batch row, column) ----------------------------------
struct NmxPowerPrepArgs {
  float* ring;                 // [cap][n_cols] raw history of the normaliser, slot = seq % cap
  const float* rows;           // [n_rows][ld] the batch (raw)
  long long ld;
  float* ext;                  // [have + n_rows][n_cols]
  double* sl;
  int have, n_rows, n_cols, cap;
  long long seq0;
};
// extended history row e: the carried rows (oldest first), then the batch; sign(x) log1p|x| next to it
NMX_DEV void nmx_power_prep_at(const NmxPowerPrepArgs& P, int e, int j) {
  if (e >= P.have + P.n_rows || j >= P.n_cols) return;
  const float v = e < P.have ? P.ring[((P.seq0 - P.have + e) % P.cap) * P.n_cols + j]
                             : P.rows[(long long)(e - P.have) * P.ld + j];
  P.ext[(long long)e * P.n_cols + j] = v;
  P.sl[(long long)e * P.n_cols + j] = nmx_pw_sl(v);
}
// after the batch: its last rows (raw) enter the ring
NMX_DEV void nmx_power_ring_at(const NmxPowerPrepArgs& P, int r, int j) {
  if (r >= P.n_rows || j >= P.n_cols || r < P.n_rows - P.cap) return;
  P.ring[((P.seq0 + r) % P.cap) * P.n_cols + j] = P.ext[((long long)P.have + r) * P.n_cols + j];
}

NMX_DEV void nmx_power_cell(const NmxPowerArgs& A, int r, int j) {
  if (r >= A.n_rows || j >= A.n_cols) return;
  if (A.colmask && !A.colmask[j]) return;
  const long long q = A.seq0 + r;
  if (q == 0) return;   // the first row ever is returned as it came (normalization.py:94-98)
  const long long hist = q < (long long)(A.cap - 1) ? q : (long long)(A.cap - 1);   // rows before the current one
  const int n = (int)hist + 1;
  const long long e0 = (long long)A.have + r - hist;   // first row of the window in the extended history
  NmxPwCol c;
  c.x = A.ext + e0 * A.n_cols + j;
  c.sl = A.sl + e0 * A.n_cols + j;
  c.stride = A.n_cols;
  c.n = n;
  double lmb, mean_t, scale_t;
  nmx_pw_fit(c, lmb, mean_t, scale_t);
  const float xr = A.ext[((long long)A.have + r) * A.n_cols + j];
  double out = (nmx_pw_transform((double)xr, lmb) - mean_t) / scale_t;   // (the current row is NOT nan_to_num'ed)
  if (A.clip > 0.f) {
    if (out < -(double)A.clip) out = -(double)A.clip;
    if (out > (double)A.clip) out = (double)A.clip;
  }
  A.rows[(long long)r * A.ld + j] = nmx_clean((float)out);
}

// nmx_k_kalman.h -- Kalman smoothing of bandpass_activity over a batch of hops.
//
// Reference: features/bandpower.py:147-163,188-189 (one filter per channel x band named in
// kalman_filter_settings.frequency_bands, predict() + update(z) per hop, feature := x[0]) with the
// model of filter/kalman_filter.py:45-78: x = [value, slope] = [0, 1], F = [[1, Tp], [0, 1]],
// H = [1, 0], R = sigma_v, Q = sigma_w^2 [[Tp^3/3, Tp^2/2], [Tp^2/2, Tp]], P0 = cov([[1,0],[0,1]])
// = [[.5, -.5], [-.5, .5]]; predict / update as in filterpy (kalman_filter_external.py:466-590,
// Joseph-form covariance update).  Float64 like the reference; one THREAD per (channel, band)
// walks the hops of the batch; state (x, P) persists in the plan between batches.
#pragma once

#include "nmx_device.h"

struct NmxKalmanArgs {
  float* out;            // [n_windows][n_outputs]
  int n_outputs, n_windows, n_channels, n_bands;
  unsigned mask;         // bit b: band b is filtered
  NmxCols cols;          // bp_cols (a = band); activity is feature slot 0
  double* state;         // [n_channels][n_bands][6] : x0, x1, P00, P01, P10, P11
  double Tp, q00, q01, q11, R;
};

NMX_DEV void nmx_kalman_init_state(double* st) {
  st[0] = 0.0; st[1] = 1.0; st[2] = 0.5; st[3] = -0.5; st[4] = -0.5; st[5] = 0.5;
}

NMX_DEV void nmx_kalman_item(const NmxKalmanArgs& A, int c, int b) {
  if (c >= A.n_channels || b >= A.n_bands || !((A.mask >> b) & 1u)) return;
  double* st = A.state + ((long long)c * A.n_bands + b) * 6;
  double x0 = st[0], x1 = st[1], p00 = st[2], p01 = st[3], p10 = st[4], p11 = st[5];
  const int col = A.cols.base + c * A.cols.ch_stride + b * A.cols.a_stride;
  const double T = A.Tp;
  for (int w = 0; w < A.n_windows; ++w) {
    float* cell = A.out + (long long)w * A.n_outputs + col;
    const double z = (double)*cell;
    // predict: x = F x ; P = F P F' + Q
    x0 = x0 + T * x1;
    const double a00 = p00 + T * p10, a01 = p01 + T * p11;   // F P
    const double n00 = a00 + a01 * T + A.q00, n01 = a01 + A.q01;
    const double n10 = p10 + p11 * T + A.q01, n11 = p11 + A.q11;
    // update: y = z - H x ; S = H P H' + R ; K = P H' / S ; x += K y ; P = (I-KH) P (I-KH)' + K R K'
    const double y = z - x0;
    const double S = n00 + A.R;
    const double k0 = n00 / S, k1 = n10 / S;
    x0 += k0 * y;
    x1 += k1 * y;
    const double i00 = 1.0 - k0, i10 = -k1;                 // I - K H = [[1-k0, 0], [-k1, 1]]
    const double b00 = i00 * n00, b01 = i00 * n01;           // (I-KH) P
    const double b10 = i10 * n00 + n10, b11 = i10 * n01 + n11;
    p00 = b00 * i00 + k0 * A.R * k0;
    p01 = b00 * i10 + b01 + k0 * A.R * k1;
    p10 = b10 * i00 + k1 * A.R * k0;
    p11 = b10 * i10 + b11 + k1 * A.R * k1;
    *cell = nmx_clean((float)x0);
  }
  st[0] = x0; st[1] = x1; st[2] = p00; st[3] = p01; st[4] = p10; st[5] = p11;
}

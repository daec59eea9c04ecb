// nmx_k_prep.h -- stream-level pre-processing kernels.
//
//  nmx_reref_tile : y[c][t] = sum_j R[c][j] * nan_to_num(x[j][t])  on the CONTINUOUS stream
//     (processing/rereference.py:99-100 after stream/data_processor.py:255).  The reference
//     re-references every window after notch/resample; all three are linear and the notch is
//     per channel, so the channel-space map commutes and is applied once per sample here
//     instead of once per (window, sample) (10x less work at 90 % overlap).  The row
//     selection data[feature_idx] is folded into R by the host.
//     Every kernel here can SUBTRACT a per-row constant on load (`sub`, NULL = none): the engine's offset split
//     (nmx_engine_dc.inc) -- the re-referenced stream then carries the signal without the rows' DC offsets, fp32
//     rounding is relative to the signal, and the constant R . sub travels next to it analytically.
//     nan_to_num comes first: a NaN sample is the VALUE 0 of the recording, i.e. -sub in the split domain.
//  nmx_nanmask_item: mask[w][j] = any(isnan(x[j][start_w : start_w + W]))
//     (stream/data_processor.py:253), one wave per (window, input row).
//  nmx_tap_item: y[w][c][0..W) = the pre-processed window the features read (the argument of
//     NMFeature.calc_feature, features/feature_processor.py:80-82), gathered from whatever layout the last
//     pre-processing stage left it in -- for user-registered host features (feature_processor.py:52-53).
#pragma once

#include "nmx_device.h"

#define NMX_REREF_ROWS 16

struct NmxRerefArgs {
  const float* x;     // [C_in][ldx]
  long long ldx;
  float* y;           // [C][ldy]
  long long ldy;
  const float* R;     // [C][C_in]
  int C, C_in;
  long long T;
  const float* sub;   // [C_in] subtracted on load, or NULL
  const float* nanv;  // [C_in] what a NaN sample becomes: the recording's value 0 in the split domain, -(offset); NULL = 0
};
NMX_DEV float nmx_clean_sub(float v, const float* sub, const float* nanv, int j) {
  if (v != v) return nanv ? nanv[j] : 0.f;
  v = nmx_clean(v);
  return sub ? v - sub[j] : v;
}

NMX_DEV float nmx_reref_store(double v) {
  // a member on the rail whose row sums to (1 + 1e-8) FLT_MAX through the rounding of the fp32 coefficients is ON the rail
  // (the reference: DBL_MAX - small = DBL_MAX), not beyond it; a true overflow exceeds it by 1 / (n - 1) of itself
  const double fmax = 3.402823466e+38;
  if (v > fmax && v < fmax * (1.0 + 1e-6)) return 3.402823466e+38f;
  if (v < -fmax && v > -fmax * (1.0 + 1e-6)) return -3.402823466e+38f;
  return (float)v;
}

// block = 256 threads <-> 256 consecutive samples; blockIdx.y <-> NMX_REREF_ROWS output rows
NMX_DEV void nmx_reref_tile(const NmxRerefArgs& A, long long t, int c0) {
  if (t >= A.T) return;
  // float64 accumulators: a 256-term fp32 dot product of samples carrying a DC offset loses ~3 digits
  // (visible in near-null spectral bins downstream); the kernel is bandwidth bound either way
  double acc[NMX_REREF_ROWS];
  for (int i = 0; i < NMX_REREF_ROWS; ++i) acc[i] = 0.0;
  const int nrow = (A.C - c0) < NMX_REREF_ROWS ? (A.C - c0) : NMX_REREF_ROWS;
  for (int j = 0; j < A.C_in; ++j) {
    const float v = nmx_clean_sub(A.x[(long long)j * A.ldx + t], A.sub, A.nanv, j);
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
    for (int i = 0; i < NMX_REREF_ROWS; ++i)
      if (i < nrow) acc[i] += (double)A.R[(long long)(c0 + i) * A.C_in + j] * (double)v;
  }
  for (int i = 0; i < nrow; ++i) A.y[(long long)(c0 + i) * A.ldy + t] = nmx_reref_store(acc[i]);
}

// Common-average style matrices R = (d - o) I + o 1 1^T (the reference's DEFAULT channel table,
// utils/channels.py:289-296, gives d = 1, o = -1/(n-1)): y_i = (d - o) x_i + o * sum_j x_j, i.e.
// one column sum per sample instead of a dense C x C product -- HBM-bound.
struct NmxCarArgs {
  const float* x;
  long long ldx;
  float* y;
  long long ldy;
  int C;
  long long T;
  float diag, off;
  const float* sub;   // [C] subtracted on load, or NULL
  const float* nanv;  // [C] a NaN sample's value (see NmxRerefArgs), or NULL
};

// float64 sums, the scale INSIDE the sum: the reference multiplies before it adds (`ref_matrix @ data`,
// processing/rereference.py:99-100), so two members at +-inf (nan_to_num: +-FLT_MAX each here, +-DBL_MAX there) leave the
// finite huge value 2 off FLT_MAX on the other channels where FLT_MAX + FLT_MAX is inf in fp32 (and off * inf = -inf on
// every channel), a pair of opposite signs cancels, and only a row whose exact value lies beyond the format's range
// comes out as inf -- on one plan, on several (sharding.py: float64 group sums from the host) and in the reference.
// The kernel waits on its loads either way (0.2 ms for 105 MB in + 105 MB out).
NMX_DEV void nmx_car_sample(const NmxCarArgs& A, long long t) {
  if (t >= A.T) return;
  const double o = (double)A.off, a = (double)A.diag - (double)A.off;
  double b = 0.0;
  for (int j = 0; j < A.C; ++j) b += o * (double)nmx_clean_sub(A.x[(long long)j * A.ldx + t], A.sub, A.nanv, j);
  for (int j = 0; j < A.C; ++j)
    A.y[(long long)j * A.ldy + t] = nmx_reref_store(a * (double)nmx_clean_sub(A.x[(long long)j * A.ldx + t], A.sub, A.nanv, j) + b);
}

#ifndef NMX_HOST_EMU
// Device form: a 256-thread workgroup owns 64 consecutive samples; wave q sums channels q, q + 4, q + 8 ...
// (a quarter of the column), the four partial sums meet in LDS, then every wave writes its own channels.
// Four times the waves of the one-thread-per-sample form and loops a quarter as long: the kernel was
// latency bound at 1.6 waves per SIMD (0.21 ms for 105 MB in + 105 MB out).
// The column sum is formed per CLASS k = channel mod 16, ascending inside a class, then ((c_q + c_(q+4)) + c_(q+8)) + c_(q+12)
// for q = 0 .. 3, then (q0 + q1) + (q2 + q3) -- whatever the number of waves: NW = 4 for a stream (a wave keeps its four
// classes in four accumulators: four independent chains), NW = 16 for the one-window call (16 workgroups in all: the 64 loop
// trips of a wave's quarter of 256 channels were 60 of that call's 450 us; here a wave owns one class).  A window and the
// stream it was cut from see the same average to the last bit.
template <int NW = 4>
NMX_DEV void nmx_car_tile(const NmxCarArgs& A, long long t0, double* red) {
  static_assert(NW == 4 || NW == 16, "waves per workgroup");
  const int lane = (int)(threadIdx.x & 63), q = (int)(threadIdx.x >> 6);
  const long long t = t0 + lane;
  const bool in = t < A.T;
  const double o = (double)A.off;
  auto term = [&](int j) { return o * (double)nmx_clean_sub(A.x[(long long)j * A.ldx + t], A.sub, A.nanv, j); };
  double s = 0.0;
  if (in) {
    if (NW == 4) {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;   // classes q, q + 4, q + 8, q + 12
      int j = q;
      for (; j + 12 < A.C; j += 16) {   // four independent loads in flight, four independent sums
        const float v0 = A.x[(long long)j * A.ldx + t], v1 = A.x[(long long)(j + 4) * A.ldx + t];
        const float v2 = A.x[(long long)(j + 8) * A.ldx + t], v3 = A.x[(long long)(j + 12) * A.ldx + t];
        if (A.sub || A.nanv) {
          c0 += o * (double)nmx_clean_sub(v0, A.sub, A.nanv, j); c1 += o * (double)nmx_clean_sub(v1, A.sub, A.nanv, j + 4);
          c2 += o * (double)nmx_clean_sub(v2, A.sub, A.nanv, j + 8); c3 += o * (double)nmx_clean_sub(v3, A.sub, A.nanv, j + 12);
        } else {
          c0 += o * (double)nmx_clean(v0); c1 += o * (double)nmx_clean(v1);
          c2 += o * (double)nmx_clean(v2); c3 += o * (double)nmx_clean(v3);
        }
      }
      if (j < A.C) c0 += term(j);
      if (j + 4 < A.C) c1 += term(j + 4);
      if (j + 8 < A.C) c2 += term(j + 8);
      s = ((c0 + c1) + c2) + c3;
    } else {
      int j = q;
      for (; j + 48 < A.C; j += 64) {   // one class: ascending, four loads in flight
        const float v0 = A.x[(long long)j * A.ldx + t], v1 = A.x[(long long)(j + 16) * A.ldx + t];
        const float v2 = A.x[(long long)(j + 32) * A.ldx + t], v3 = A.x[(long long)(j + 48) * A.ldx + t];
        s += o * (double)nmx_clean_sub(v0, A.sub, A.nanv, j); s += o * (double)nmx_clean_sub(v1, A.sub, A.nanv, j + 16);
        s += o * (double)nmx_clean_sub(v2, A.sub, A.nanv, j + 32); s += o * (double)nmx_clean_sub(v3, A.sub, A.nanv, j + 48);
      }
      for (; j < A.C; j += 16) s += term(j);
    }
  }
  red[q * 64 + lane] = s;
  __syncthreads();
  if (!in) return;
  double b;
  if (NW == 4) {
    b = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
  } else {
    double p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      p[k] = ((red[k * 64 + lane] + red[(k + 4) * 64 + lane]) + red[(k + 8) * 64 + lane]) + red[(k + 12) * 64 + lane];
    b = (p[0] + p[1]) + (p[2] + p[3]);
  }
  const double a = (double)A.diag - (double)A.off;
  for (int j = q; j < A.C; j += NW)
    A.y[(long long)j * A.ldy + t] = nmx_reref_store(a * (double)nmx_clean_sub(A.x[(long long)j * A.ldx + t], A.sub, A.nanv, j) + b);
}
#endif

// Structured re-reference matrices: every row is a few explicit taps plus a multiple of one GROUP SUM,
//   y_r = sum_{s < 4} coef[r][s] * x[idx[r][s]]  +  b_r * S_{g_r},   S_g = sum_{j in G_g} x_j
// which covers what processing/rereference.py:52-86 can build: "average" rows (all other good channels
// of the same type, :61-63 -> G = the type group, tap (1 - o) on the channel itself), bipolar rows
// (:65-79 -> 2-3 taps, no group) and, with the channel pick folded in, ROW SUBSETS of such matrices
// (a channel shard of a jointly referenced array: every GPU reads the group's rows once per sample
// instead of running a dense [C x C_in] product).  The host finds the structure (nmx_engine.inc).
#define NMX_RS_TAPS 4
#define NMX_RS_GROUPS 4
struct NmxRerefStructArgs {
  const float* x;        // [C_in][ldx]
  long long ldx;
  float* y;              // [C][ldy]
  long long ldy;
  int C, C_in;
  long long T;
  const int* row_idx;    // [C][NMX_RS_TAPS] (unused taps: idx 0, coef 0)
  const float* row_coef; // [C][NMX_RS_TAPS]
  const int* row_group;  // [C] group or -1
  const float* row_b;    // [C]
  int n_groups;
  const int* members;    // concatenated member rows of the groups
  int group_off[NMX_RS_GROUPS + 1];
  const float* sub;      // [C_in] subtracted on load, or NULL
  const float* nanv;     // [C_in] a NaN sample's value (see NmxRerefArgs), or NULL
};

NMX_DEV void nmx_reref_struct_sample(const NmxRerefStructArgs& A, long long t) {
  if (t >= A.T) return;
  double S[NMX_RS_GROUPS];
  for (int g = 0; g < A.n_groups; ++g) {
    double s = 0.0;
    for (int m = A.group_off[g]; m < A.group_off[g + 1]; ++m)
      s += (double)nmx_clean_sub(A.x[(long long)A.members[m] * A.ldx + t], A.sub, A.nanv, A.members[m]);
    S[g] = s;
  }
  for (int r = 0; r < A.C; ++r) {
    double acc = A.row_group[r] >= 0 ? (double)A.row_b[r] * S[A.row_group[r]] : 0.0;
    for (int k = 0; k < NMX_RS_TAPS; ++k) {
      const float c = A.row_coef[r * NMX_RS_TAPS + k];
      if (c != 0.f) acc += (double)c * (double)nmx_clean_sub(A.x[(long long)A.row_idx[r * NMX_RS_TAPS + k] * A.ldx + t], A.sub, A.nanv, A.row_idx[r * NMX_RS_TAPS + k]);
    }
    A.y[(long long)r * A.ldy + t] = nmx_reref_store(acc);
  }
}

#ifndef NMX_HOST_EMU
// Device form (as nmx_car_tile): a 256-thread workgroup owns 64 consecutive samples; wave q sums the members
// q, q + 4, ... of every group (float64 partial sums: up to thousands of DC-carrying terms), the partials
// meet in LDS, then wave q writes the rows q, q + 4, ...  `red` holds NMX_RS_GROUPS * 256 doubles.
NMX_DEV void nmx_reref_struct_tile(const NmxRerefStructArgs& A, long long t0, double* red) {
  const int lane = (int)(threadIdx.x & 63), q = (int)(threadIdx.x >> 6);
  const long long t = t0 + lane;
  const bool in = t < A.T;
  for (int g = 0; g < A.n_groups; ++g) {
    double s = 0.0;
    if (in) {
      int m = A.group_off[g] + q;
      const int end = A.group_off[g + 1];
      for (; m + 12 < end; m += 16) {   // four independent loads in flight
        const int j0 = A.members[m], j1 = A.members[m + 4], j2 = A.members[m + 8], j3 = A.members[m + 12];
        const float v0 = A.x[(long long)j0 * A.ldx + t], v1 = A.x[(long long)j1 * A.ldx + t];
        const float v2 = A.x[(long long)j2 * A.ldx + t], v3 = A.x[(long long)j3 * A.ldx + t];
        s += ((double)nmx_clean_sub(v0, A.sub, A.nanv, j0) + (double)nmx_clean_sub(v1, A.sub, A.nanv, j1)) +
             ((double)nmx_clean_sub(v2, A.sub, A.nanv, j2) + (double)nmx_clean_sub(v3, A.sub, A.nanv, j3));
      }
      for (; m < end; m += 4) s += (double)nmx_clean_sub(A.x[(long long)A.members[m] * A.ldx + t], A.sub, A.nanv, A.members[m]);
    }
    red[(g * 4 + q) * 64 + lane] = s;
  }
  __syncthreads();
  if (!in) return;
  double S[NMX_RS_GROUPS];
  for (int g = 0; g < A.n_groups; ++g)
    S[g] = (red[(g * 4) * 64 + lane] + red[(g * 4 + 1) * 64 + lane]) + (red[(g * 4 + 2) * 64 + lane] + red[(g * 4 + 3) * 64 + lane]);
  for (int r = q; r < A.C; r += 4) {
    const int g = A.row_group[r];
    double acc = 0.0;
    for (int k = 0; k < NMX_RS_GROUPS; ++k) if (k == g) acc = (double)A.row_b[r] * S[k];   // (no dynamic register indexing)
    for (int k = 0; k < NMX_RS_TAPS; ++k) {
      const float c = A.row_coef[r * NMX_RS_TAPS + k];
      if (c != 0.f) acc += (double)c * (double)nmx_clean_sub(A.x[(long long)A.row_idx[r * NMX_RS_TAPS + k] * A.ldx + t], A.sub, A.nanv, A.row_idx[r * NMX_RS_TAPS + k]);
    }
    A.y[(long long)r * A.ldy + t] = nmx_reref_store(acc);
  }
}
#endif

// The split without a re-reference in front (a notch reads the recording's rows directly): y[c][t] = nan_to_num(x[c][t]) -
// sub[c] on the sample range of a chunk, once per sample like the re-reference kernels (nmx_engine_dc.inc).
struct NmxShiftArgs {
  const float* x;
  long long ldx;
  float* y;
  long long ldy;
  int C;
  long long T;
  const float* sub;
  const float* nanv;
};
NMX_DEV void nmx_shift_sample(const NmxShiftArgs& A, long long t, int c) {
  if (t < A.T) A.y[(long long)c * A.ldy + t] = nmx_clean_sub(A.x[(long long)c * A.ldx + t], A.sub, A.nanv, c);
}

// ---- the stand-alone ReReferencer in float64 (nmx_reref_f64): y = R x, one thread per sample column and NMX_REREF64_ROWS
// output rows; x is read once per row group (coalesced 8-byte loads), R broadcast from the scalar cache ----------------------
#define NMX_REREF64_ROWS 8
struct NmxReref64Args {
  const double* x;   // [C_in][ldx]
  long long ldx;
  double* y;         // [C][ldy]
  long long ldy;
  const double* R;   // [C][C_in]
  int C, C_in;
  long long T;
};
NMX_DEV void nmx_reref64_tile(const NmxReref64Args& A, long long t, int c0) {
  if (t >= A.T) return;
  double acc[NMX_REREF64_ROWS];
  for (int i = 0; i < NMX_REREF64_ROWS; ++i) acc[i] = 0.0;
  const int nrow = (A.C - c0) < NMX_REREF64_ROWS ? (A.C - c0) : NMX_REREF64_ROWS;
  for (int j = 0; j < A.C_in; ++j) {
    const double v = A.x[(long long)j * A.ldx + t];
#ifndef NMX_HOST_EMU
#pragma unroll
#endif
    for (int i = 0; i < NMX_REREF64_ROWS; ++i)
      if (i < nrow) acc[i] += A.R[(long long)(c0 + i) * A.C_in + j] * v;
  }
  for (int i = 0; i < nrow; ++i) A.y[(long long)(c0 + i) * A.ldy + t] = acc[i];
}

struct NmxNanMaskArgs {
  const float* x;
  long long ldx;
  const long long* starts;
  unsigned char* mask;   // [n_windows][C_in]
  int C_in, W;
};

NMX_DEV void nmx_nanmask_item(const NmxNanMaskArgs& A, int w, int j, float* smem) {
  const float* src = A.x + (long long)j * A.ldx + A.starts[w];
  int any = 0;
  for (int i = NMX_TID; i < A.W; i += NMX_NT) {
    const float v = src[i];
    any |= (v != v);
  }
  any = nmx_block_or(any, smem);
  if (NMX_TID == 0) A.mask[(long long)w * A.C_in + j] = (unsigned char)(any ? 1 : 0);
}

struct NmxTapArgs {
  const float* x;           // last pre-processing stage's output (or the recording itself)
  long long ch_stride, win_stride;
  const long long* starts;  // per-window start sample or NULL
  float* y;                 // [n_windows][C][W]
  int C, W;
  int clean;                // nan_to_num while copying (no stage has cleaned the samples yet)
  const float* add;         // [C] constant added to every sample of a row (the carried offset: a kernel that cannot
                            // take it on load reads this copy instead), or NULL
};

NMX_DEV void nmx_tap_item(const NmxTapArgs& A, int w, int c) {
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + (A.starts ? A.starts[w] : 0);
  float* dst = A.y + ((long long)w * A.C + c) * A.W;
  for (int i = NMX_TID; i < A.W; i += NMX_NT) {
    float v = src[i];
    v = A.clean ? nmx_clean(v) : v;
    dst[i] = A.add ? v + A.add[c] : v;
  }
}

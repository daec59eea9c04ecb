// nmx_k_sharpwave.h -- kernel E: SharpwaveAnalyzer.analyze_waveform + estimators
// (features/sharpwaves.py:259-328,330-465) for one pre-filtered series per WAVE.
//
// Work per item ((window, channel, filter), both polarities): local maxima with SciPy's
// plateau rule, SciPy's priority-ordered `distance` suppression (greedy by height, solved
// here as a parallel fixed point -- a peak is kept once every higher neighbour within
// `distance` is removed, removed once a higher neighbour is kept), trough <-> peak pairing,
// the 13 per-trough quantities and the estimators.  Everything lives in LDS; ordered
// compaction uses contiguous per-lane chunks + a wave prefix sum so results are
// deterministic.  Ties of equal height inside `distance`: the later peak wins (SciPy's own
// order there is NumPy's unstable argsort, i.e. implementation-defined).
#pragma once

#include "nmx_device.h"

#include "../../include/nmx.h"  // NMX_SW_* feature ids and NMX_SWE_* estimator ids

struct NmxSharpArgs {
  const float* y;   // [n_windows][C][n_filters][W] pre-filtered series
  float* out;
  int n_outputs, n_windows, n_channels, n_filters, W;
  float ms;         // 1000 / sfreq
  int sharp_off;    // int(5 * (1000 / sfreq))
  int dist_peaks, dist_troughs;   // ceil(distance), samples
  int est_peaks, est_troughs;     // polarity enabled flags
  int between;
  int n_combos;
  int combo_feature[NMX_MAX_SW_COMBOS_DEV], combo_est[NMX_MAX_SW_COMBOS_DEV];
  int combo_slot[NMX_MAX_SW_COMBOS_DEV];   // output slot or -1 (num_peaks when between)
  unsigned feature_mask;          // bit f: feature f used by some combo
  NmxCols cols;       // a = filter, b = slot (x n_polarities + polarity when !between)
  NmxCols np_cols;    // num_peaks (between mode), a = filter
  int has_num_peaks;
  int off_z, off_pk, off_tr, off_st, off_lf, off_rt, off_vals, off_res, off_red, lds_floats;
};

#ifdef NMX_HOST_EMU
NMX_DEV int nmx_wave_excl_sum_i(int v, int* total) { *total = v; return 0; }
#else
NMX_DEV int nmx_wave_excl_sum_i(int v, int* total) {
  const int lane = threadIdx.x & 63;
  int inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  *total = __shfl(inc, 63);
  return inc - v;
}
#endif

// local maxima of sgn * z (SciPy _local_maxima_1d), ordered, into pos[]; returns count
NMX_DEV int nmx_local_maxima(const float* z, float sgn, int W, int* pos) {
  const int n_idx = W - 2;  // candidate indices 1 .. W-2
  const int chunk = n_idx > 0 ? (n_idx + NMX_NT - 1) / NMX_NT : 0;
  const int i0 = 1 + NMX_TID * chunk;
  const int i1 = (i0 + chunk) < (W - 1) ? (i0 + chunk) : (W - 1);
  int cnt = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int base = 0;
    if (pass == 1) {
      int total;
      base = nmx_wave_excl_sum_i(cnt, &total);
      cnt = total;
    }
    int k = 0;
    for (int i = i0; i < i1; ++i) {
      const float v = sgn * z[i];
      if (sgn * z[i - 1] < v) {
        int ahead = i + 1;
        while (ahead < W - 1 && sgn * z[ahead] == v) ++ahead;
        if (sgn * z[ahead] < v) {
          if (pass == 1) pos[base + k] = (i + ahead - 1) >> 1;
          ++k;
        }
      }
    }
    if (pass == 0) cnt = k;
  }
  NMX_SYNC();
  return cnt;
}

// SciPy _select_by_peak_distance: in-place state st[] (1 keep, 2 removed), then ordered
// compaction of pos[] ; returns the new count
NMX_DEV int nmx_select_by_distance(const float* z, float sgn, int* pos, int n, int dist, int* st,
                                   float* red) {
  if (dist <= 1 || n <= 1) return n;
  for (int j = NMX_TID; j < n; j += NMX_NT) st[j] = 0;
  NMX_SYNC();
  for (;;) {
    int undecided = 0;
    for (int j = NMX_TID; j < n; j += NMX_NT) {
      if (st[j] != 0) continue;
      const int pj = pos[j];
      const float vj = sgn * z[pj];
      bool removed = false, wait = false;
      for (int k = j - 1; k >= 0 && pj - pos[k] < dist; --k) {
        const float vk = sgn * z[pos[k]];
        if (vk > vj) {  // strictly higher; equal height: the later index (j) has priority
          const int s = st[k];
          if (s == 1) removed = true; else if (s == 0) wait = true;
        }
      }
      for (int k = j + 1; k < n && pos[k] - pj < dist; ++k) {
        const float vk = sgn * z[pos[k]];
        if (vk >= vj) {
          const int s = st[k];
          if (s == 1) removed = true; else if (s == 0) wait = true;
        }
      }
      if (removed) st[j] = 2;
      else if (!wait) st[j] = 1;
      else undecided = 1;
    }
    NMX_SYNC();
    if (!nmx_block_or(undecided, red)) break;
  }
  // ordered compaction of kept peaks (in place is safe: write index <= read index, but lanes
  // run concurrently -> stage through registers per chunk)
  const int chunk = (n + NMX_NT - 1) / NMX_NT;
  const int i0 = NMX_TID * chunk, i1 = (i0 + chunk) < n ? (i0 + chunk) : n;
  int cnt = 0;
  for (int i = i0; i < i1; ++i) cnt += (st[i] == 1);
  int total;
  const int base = nmx_wave_excl_sum_i(cnt, &total);
  // second array needed: reuse st as destination after reading my chunk's kept positions
  int k = 0;
  NMX_SYNC();
  // two-step: first write compacted positions into st-sized scratch held in registers is not
  // possible for arbitrary chunk sizes; encode instead: st[i] = kept ? pos[i] : -1, then gather
  for (int i = i0; i < i1; ++i) st[i] = (st[i] == 1) ? pos[i] : -1;
  NMX_SYNC();
  for (int i = i0; i < i1; ++i)
    if (st[i] >= 0) { pos[base + k] = st[i]; ++k; }
  NMX_SYNC();
  return total;
}

NMX_DEV float nmx_sw_estimate(int est, const float* v, int n, float* red) {
  if (n == 0) return 0.f;  // sharpwaves.py:294: empty -> 0
  switch (est) {
    case NMX_SWE_MEAN: return nmx_est_mean(v, n, red);
    case NMX_SWE_MEDIAN: return nmx_est_median(v, n, red);
    case NMX_SWE_MAX: return nmx_est_max(v, n, red);
    case NMX_SWE_MIN: return nmx_est_min(v, n, red);
    default: {
      const float m = nmx_est_mean(v, n, red);
      const float s = nmx_est_std(v, n, m, red);
      return s * s;
    }
  }
}

NMX_DEV float nmx_sw_pair(int est, float a, float b) {
  switch (est) {
    case NMX_SWE_MEAN:
    case NMX_SWE_MEDIAN: return 0.5f * (a + b);
    case NMX_SWE_MAX: return nmx_nanmax(a, b);
    case NMX_SWE_MIN: return nmx_nanmin(a, b);
    default: { const float m = 0.5f * (a + b); return 0.5f * ((a - m) * (a - m) + (b - m) * (b - m)); }
  }
}

// one WAVE per (window, channel, filter)
NMX_DEV void nmx_sharp_item(const NmxSharpArgs& A, int w, int c, int fi, float* smem) {
  float* z = smem + A.off_z;
  int* pk = (int*)(smem + A.off_pk);
  int* tr = (int*)(smem + A.off_tr);
  int* st = (int*)(smem + A.off_st);
  int* lf = (int*)(smem + A.off_lf);
  int* rt = (int*)(smem + A.off_rt);
  float* vals = smem + A.off_vals;
  float* res = smem + A.off_res;   // [2][n_combos] + [2] num_peaks
  float* red = smem + A.off_red;
  const int W = A.W;
  const float* src = A.y + (((long long)w * A.n_channels + c) * A.n_filters + fi) * W;
  for (int i = NMX_TID; i < W; i += NMX_NT) z[i] = src[i];
  NMX_SYNC();
  float* row = A.out + (long long)w * A.n_outputs;
  int pol_slot = 0;
  const int n_pol = (A.est_peaks ? 1 : 0) + (A.est_troughs ? 1 : 0);
  for (int pol = 0; pol < 2; ++pol) {
    if ((pol == 0 && !A.est_peaks) || (pol == 1 && !A.est_troughs)) continue;
    const float sgn = pol == 0 ? 1.f : -1.f;   // "Trough" analysis runs on -y
    // peaks of sgn*z (distance_peaks) and troughs = peaks of -sgn*z (distance_troughs)
    int nPk = nmx_local_maxima(z, sgn, W, pk);
    nPk = nmx_select_by_distance(z, sgn, pk, nPk, A.dist_peaks, st, red);
    int nTr = nmx_local_maxima(z, -sgn, W, tr);
    nTr = nmx_select_by_distance(z, -sgn, tr, nTr, A.dist_troughs, st, red);
    // pairing (sharpwaves.py:347-374)
    int n_leftinv = 0, lastv = 0, n_pairs = 0;
    for (int i = NMX_TID; i < nTr; i += NMX_NT) {
      const int t = tr[i];
      int lo = 0, hi = nPk;   // first peak with pos >= t
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (pk[mid] < t) lo = mid + 1; else hi = mid; }
      st[i] = lo;
      if (lo == 0) ++n_leftinv;
      else if (lo < nPk) { lastv = i > lastv ? i : lastv; ++n_pairs; }
    }
    n_leftinv = nmx_block_sum_i(n_leftinv, red);
    n_pairs = nmx_block_sum_i(n_pairs, red);
    lastv = (int)nmx_block_max((float)lastv, red);
    NMX_SYNC();
    const int first_valid = n_leftinv;
    int last_excl = (lastv + 1) < nTr ? (lastv + 1) : nTr;
    const int nT = last_excl > first_valid ? last_excl - first_valid : 0;
    const int* trv = tr + first_valid;  // trough list after the reference's slice
    for (int p = NMX_TID; p < n_pairs; p += NMX_NT) {
      const int ptr = st[first_valid + p];
      lf[p] = pk[ptr - 1];
      rt[p] = pk[ptr];
    }
    NMX_SYNC();
    const int nPT = (n_pairs == nT) ? n_pairs : 0;  // arrays that broadcast pairs with troughs
    if (NMX_TID == 0) res[2 * A.n_combos + pol] = (float)nT;
    for (int f = 0; f < NMX_SW_NFEAT; ++f) {
      if (!(A.feature_mask & (1u << f)) || f == NMX_SW_NUM_PEAKS) continue;
      int n = 0;
      NMX_SYNC();
      switch (f) {
        case NMX_SW_PEAK_LEFT: n = n_pairs; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = sgn * z[lf[p]]; break;
        case NMX_SW_PEAK_RIGHT: n = n_pairs; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = sgn * z[rt[p]]; break;
        case NMX_SW_TROUGH: n = nT; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = sgn * z[trv[p]]; break;
        case NMX_SW_WIDTH: n = n_pairs; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = (float)(rt[p] - lf[p]); break;
        case NMX_SW_PROMINENCE:
          n = nPT;
          for (int p = NMX_TID; p < n; p += NMX_NT)
            vals[p] = fabsf((sgn * z[rt[p]] + sgn * z[lf[p]]) * 0.5f - sgn * z[trv[p]]);
          break;
        case NMX_SW_INTERVAL:
          n = nT;
          for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = p == 0 ? 0.f : (float)(trv[p] - trv[p - 1]) * A.ms;
          break;
        case NMX_SW_DECAY_TIME: n = nPT; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = (float)(lf[p] - trv[p]) * A.ms; break;
        case NMX_SW_RISE_TIME: n = nPT; for (int p = NMX_TID; p < n; p += NMX_NT) vals[p] = (float)(rt[p] - trv[p]) * A.ms; break;
        case NMX_SW_SHARPNESS: {
          // ordered compaction of troughs with a +-sharp_off margin (sharpwaves.py:393-406)
          const int s = A.sharp_off;
          const int chunk = (nT + NMX_NT - 1) / NMX_NT;
          const int i0 = NMX_TID * chunk, i1 = (i0 + chunk) < nT ? (i0 + chunk) : nT;
          int cnt = 0;
          for (int i = i0; i < i1; ++i) cnt += (trv[i] - s > 0 && trv[i] + s < W);
          int total;
          const int base = nmx_wave_excl_sum_i(cnt, &total);
          int k = 0;
          for (int i = i0; i < i1; ++i) {
            const int t = trv[i];
            if (t - s > 0 && t + s < W) {
              vals[base + k] = sgn * z[t] - 0.5f * (sgn * z[t - s] + sgn * z[t + s]);
              ++k;
            }
          }
          n = total;
          break;
        }
        case NMX_SW_RISE_STEEPNESS:
        case NMX_SW_DECAY_STEEPNESS:
        case NMX_SW_SLOPE_RATIO:
          n = nPT;
          for (int p = NMX_TID; p < n; p += NMX_NT) {
            float rise = 0.f, decay = 0.f;
            for (int j = lf[p]; j <= trv[p]; ++j) {
              const float d = j > 0 ? fabsf(z[j] - z[j - 1]) : 0.f;
              rise = d > rise ? d : rise;
            }
            for (int j = trv[p]; j <= rt[p]; ++j) {
              const float d = j > 0 ? fabsf(z[j] - z[j - 1]) : 0.f;
              decay = d > decay ? d : decay;
            }
            vals[p] = f == NMX_SW_RISE_STEEPNESS ? rise : (f == NMX_SW_DECAY_STEEPNESS ? decay : rise - decay);
          }
          break;
        default: break;
      }
      NMX_SYNC();
      for (int cb = 0; cb < A.n_combos; ++cb) {
        if (A.combo_feature[cb] != f) continue;
        const float r = nmx_sw_estimate(A.combo_est[cb], vals, n, red);
        if (NMX_TID == 0) res[pol * A.n_combos + cb] = r;
      }
    }
    NMX_SYNC();
    if (!A.between && NMX_TID == 0) {
      for (int cb = 0; cb < A.n_combos; ++cb) {
        if (A.combo_slot[cb] < 0) continue;
        const float v = A.combo_feature[cb] == NMX_SW_NUM_PEAKS ? res[2 * A.n_combos + pol]
                                                                : res[pol * A.n_combos + cb];
        row[A.cols.base + c * A.cols.ch_stride + fi * A.cols.a_stride +
            A.combo_slot[cb] * A.cols.b_stride + pol_slot] = v;
      }
    }
    ++pol_slot;
  }
  NMX_SYNC();
  if (A.between && NMX_TID == 0 && n_pol == 2) {
    for (int cb = 0; cb < A.n_combos; ++cb) {
      if (A.combo_slot[cb] < 0) continue;
      row[A.cols.base + c * A.cols.ch_stride + fi * A.cols.a_stride + A.combo_slot[cb] * A.cols.b_stride] =
          nmx_sw_pair(A.combo_est[cb], res[cb], res[A.n_combos + cb]);
    }
    if (A.has_num_peaks)
      row[A.np_cols.base + c * A.np_cols.ch_stride + fi * A.np_cols.a_stride] =
          0.5f * (res[2 * A.n_combos] + res[2 * A.n_combos + 1]);
  }
}

// nmx_k_scan.h -- register-resident time-domain scan: Hjorth (hjorth_raw.py:24-42), Raw (:51-57) and
// LineLength (linelength.py:11-21) of one (window, channel) per WAVE, no LDS.
//
// This is the HBM-bound member of the path (4 W bytes in, a few floats out, ~15 flop per
// sample).  The generic kernel (nmx_k_timeosc.h) stages the window in LDS for the transforms that
// follow; when NO oscillatory feature is enabled this kernel is used instead: every lane loads up to
// four 16-byte groups of the window through a buffer descriptor (range-checked, so the ragged end needs
// no branches), keeps them in registers for both passes of the two-pass variances, and gets the two
// samples that follow each group from the next lane with DPP wave shifts.  Reductions on the DPP path.
// W <= 1024 (64 lanes x 4 groups x 4 samples).  Same formulas as nmx_time_osc_item.
#pragma once

#include "nmx_k_timeosc.h"

// NmxTimeOscArgs::dcf[c] for a wave-uniform c: through the CONSTANT address space on the device (an s_load, lgkmcnt) --
// a vector load's s_waitcnt vmcnt(0) would also wait for every load in flight (the prefetching kernels); the table is
// written by the host before the launch
NMX_DEV float nmx_dc_of(const NmxTimeOscArgs& A, int c) {
  if (!A.dcf) return 0.f;
#ifdef NMX_HOST_EMU
  return A.dcf[c];
#else
  typedef const float __attribute__((address_space(4)))* nmx_cf_p;
  return ((nmx_cf_p)(unsigned long long)A.dcf)[c];
#endif
}

#ifndef NMX_HOST_EMU
// value of `v` in lane + 1; lane 63 receives `wrap` (wave-uniform)
NMX_DEV float nmx_from_next_lane(float v, float wrap, int lane) {
  const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));   // wave_shl:1
  return lane == 63 ? wrap : t;
}

typedef float nmx_f4 __attribute__((ext_vector_type(4)));

// group k of a lane = samples 4 (lane + 64 k) .. + 3 ; out-of-range dwords read 0
struct NmxScanRegs {
  float x[4][6];   // [group][0..3 own samples, 4..5 the two samples that follow]
  float sum;       // sum of the window (set by nmx_scan_emit)
};

// window -> registers (w, c wave-uniform)
NMX_DEV void nmx_scan_load(const NmxTimeOscArgs& A, int w, int c, NmxScanRegs& R) {
  const int lane = (int)(threadIdx.x & 63);
  const int W = A.W;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride +
                     (A.starts ? nmx_uniform_ll(A.starts[w]) : 0ll);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4 * W, 0x00020000);
  typedef nmx_f4 f4;
  float (&x)[4][6] = R.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f4 v;
    const int off = 16 * lane + 1024 * k;
    if ((W & 3) == 0) {   // every group is entirely inside or outside the row: one 16-byte load
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      v.x = __uint_as_float(r.x); v.y = __uint_as_float(r.y); v.z = __uint_as_float(r.z); v.w = __uint_as_float(r.w);
    } else {              // dword accesses are range-checked one by one
      v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
      v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + 4, 0, 0));
      v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + 8, 0, 0));
      v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + 12, 0, 0));
    }
    if (A.clean_on_load) { v.x = nmx_clean_bl(v.x); v.y = nmx_clean_bl(v.y); v.z = nmx_clean_bl(v.z); v.w = nmx_clean_bl(v.w); }
    x[k][0] = v.x; x[k][1] = v.y; x[k][2] = v.z; x[k][3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // lane 63's successor is lane 0 of the next group
    const float w0 = k < 3 ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x[k + 1][0]))) : 0.f;
    const float w1 = k < 3 ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x[k + 1][1]))) : 0.f;
    x[k][4] = nmx_from_next_lane(x[k][0], w0, lane);
    x[k][5] = nmx_from_next_lane(x[k][1], w1, lane);
  }
}

// Hjorth / LineLength / Raw of the window held in R -> out (same formulas as nmx_time_osc_item)
NMX_DEV void nmx_scan_emit(const NmxTimeOscArgs& A, int w, int c, NmxScanRegs& R) {
  const int lane = (int)(threadIdx.x & 63);
  const int W = A.W;
  float (&x)[4][6] = R.x;
  // pass 1: sums of x, dx, d2x and |dx|
  // a group whose last lane still has two successors inside the window needs no masks (wave-uniform
  // test); for W = 1000 that is three of the four groups
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n0 = 4 * (lane + 64 * k);
    if (4 * (63 + 64 * k) + 5 < W) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = x[k][j], x1 = x[k][j + 1], x2 = x[k][j + 2];
        const float d1 = x1 - x0;
        p0 += x0; p1 += d1; p3 += fabsf(d1); p2 += (x2 - x1) - d1;
      }
    } else if (256 * k < W) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        const float x0 = x[k][j], x1 = x[k][j + 1], x2 = x[k][j + 2];
        const float d1 = x1 - x0;
        p0 += n < W ? x0 : 0.f;
        p1 += n + 1 < W ? d1 : 0.f;
        p3 += n + 1 < W ? fabsf(d1) : 0.f;
        p2 += n + 2 < W ? (x2 - x1) - d1 : 0.f;
      }
    }
  }
  auto add = [](float a, float b) { return a + b; };
  p0 = nmx_wave_reduce(p0, 0.f, add); p1 = nmx_wave_reduce(p1, 0.f, add);
  p2 = nmx_wave_reduce(p2, 0.f, add); p3 = nmx_wave_reduce(p3, 0.f, add);
  R.sum = p0;
  const float m0 = p0 / (float)W, m1 = p1 / (float)(W - 1), m2 = p2 / (float)(W - 2);
  // pass 2: mean-shifted sums of squares from the same registers
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n0 = 4 * (lane + 64 * k);
    if (4 * (63 + 64 * k) + 5 < W) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = x[k][j], x1 = x[k][j + 1], x2 = x[k][j + 2];
        const float d1 = x1 - x0;
        const float e0 = x0 - m0, e1 = d1 - m1, e2 = ((x2 - x1) - d1) - m2;
        q0 += e0 * e0; q1 += e1 * e1; q2 += e2 * e2;
      }
    } else if (256 * k < W) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        const float x0 = x[k][j], x1 = x[k][j + 1], x2 = x[k][j + 2];
        const float d1 = x1 - x0;
        const float e0 = x0 - m0, e1 = d1 - m1, e2 = ((x2 - x1) - d1) - m2;
        q0 += n < W ? e0 * e0 : 0.f;
        q1 += n + 1 < W ? e1 * e1 : 0.f;
        q2 += n + 2 < W ? e2 * e2 : 0.f;
      }
    }
  }
  q0 = nmx_wave_reduce(q0, 0.f, add); q1 = nmx_wave_reduce(q1, 0.f, add); q2 = nmx_wave_reduce(q2, 0.f, add);
  float* out_row = A.out + (long long)w * A.n_outputs;
  // last sample of the window (Raw): element W - 1 lives in group (W - 1) / 256, lane ((W - 1) / 4) % 64
  const int gl = ((W - 1) >> 2) & 63, gk = (W - 1) >> 8, gj = (W - 1) & 3;
  float last = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k == gk && j == gj) last = x[k][j];
  last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, last), gl));
  if (lane == 0) {
    if (A.features & NMXD_F_HJORTH) {
      const float v0 = q0 / (float)W, v1 = q1 / (float)(W - 1), v2 = q2 / (float)(W - 2);
      const float mob = nmx_nan_to_num(sqrtf(v1 / v0));
      const float comp = nmx_nan_to_num(sqrtf(v2 / v1) / mob);
      const int col = A.hjorth_cols.base + c * A.hjorth_cols.ch_stride;
      out_row[col] = nmx_nan_to_num(v0);
      out_row[col + A.hjorth_cols.a_stride] = mob;
      out_row[col + 2 * A.hjorth_cols.a_stride] = comp;
    }
    if (A.features & NMXD_F_LINELENGTH) {
      const float wm1 = (float)(W - 1);
      out_row[A.ll_cols.base + c * A.ll_cols.ch_stride] = p3 / wm1 / wm1;
    }
    if (A.features & NMXD_F_RAW) out_row[A.raw_cols.base + c * A.raw_cols.ch_stride] = last + nmx_dc_of(A, c);
  }
}

NMX_DEV void nmx_scan_item(const NmxTimeOscArgs& A, int w, int c) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  NmxScanRegs R;
  nmx_scan_load(A, w, c, R);
  nmx_scan_emit(A, w, c, R);
}
#endif

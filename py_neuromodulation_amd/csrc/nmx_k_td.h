// nmx_k_td.h -- packed time-domain statistics of one (window, channel) per WAVE: Hjorth (hjorth_raw.py:24-42),
// Raw (:51-57), LineLength (linelength.py:11-21) and the window sum / centred window the FFT path needs.
//
// This is what made the time / oscillatory kernel issue bound: the round-2 formulation (nmx_k_scan.h) spent 596 VALU
// instructions per item on it -- two thirds of the FFT + Hjorth + LineLength item -- on scalar fp32 operations, a
// nan_to_num per sample, DPP shuffles for the successors of a lane's samples and nine IEEE divisions.  Here:
//   * a lane's group k holds samples n = 4 (lane + 64 k) + 0..3 as TWO 64-bit register pairs and loads its two
//     successors (x4, x5) itself with one extra 8-byte buffer load (range-checked: 0 beyond the window) -- no
//     cross-lane traffic at all;
//   * every per-sample operation is PACKED (v_pk_add / v_pk_fma / v_pk_mul_f32: two samples per instruction).  The
//     first difference of the aligned pair (x0, x1) needs the UNALIGNED pair (x1, x2): one v_pk_mov per pair builds
//     Y = x shifted by one; Z = x shifted by two is the next aligned pair, free:
//         d1 = Y - X,   d2 = Z - 2 Y + X;
//   * the sums of d1 and d2 telescope (x[W-1] - x[0], d1[W-2] - d1[0]) -- read from the ends of the row, not summed;
//   * nan_to_num is NOT applied per sample: a NaN or an infinity anywhere in the window makes the window sum
//     non-finite, which the wave notices after the first reduction and then takes the round-2 path (cleaning
//     loads, scalar arithmetic) for that item.  Finite data -- the only data a recording normally holds -- never pays;
//   * ragged tail (W = 1000: group 3 ends at lane 57): the three validity masks (x: n < W, d1: n + 1 < W,
//     d2: n + 2 < W) multiply the deviations as packed 0 / 1 factors, only in the groups that are not full;
//   * the last divisions and square roots run on v_rcp_f32 / v_sqrt_f32 (1 ulp) when every variance is a normal
//     positive number, on the IEEE sequences (with the reference's nan_to_num placement) otherwise.
// 200 VALU instructions per item instead of 596; results agree with the round-2 path to fp32 rounding (the
// parity tests compare both with the float64 oracle).  Needs W % 4 == 0, 8 <= W <= 1024.
#pragma once

#include "nmx_k_scan.h"

#ifndef NMX_HOST_EMU

struct NmxTdRegs {
  nmx_f4 x[4];      // group k: x0..x3
  nmx_c2 s[4];      // group k: the two samples that follow (x4, x5)
  nmx_c2 first;     // (x[0], x[1])
  nmx_c2 last;      // (x[W-2], x[W-1])
  float sum;        // sum of the window
};

NMX_DEV bool nmx_td_ok(const NmxTimeOscArgs& A) { return (A.W & 3) == 0 && A.W >= 8 && A.W <= 1024; }

NMX_DEV nmx_c2 nmx_td_ldc2(const __amdgpu_buffer_rsrc_t rs, int off) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
  return nmx_mk2(__uint_as_float(r.x), __uint_as_float(r.y));
}

// (a.y, b.x): a pair that straddles two aligned register pairs, assembled by ONE v_pk_mov_b32 (the compiler builds it
// from two v_mov_b32 -- and an instruction that only moves data costs the issue slot of one that computes)
NMX_DEV nmx_c2 nmx_pk_hilo(nmx_c2 a, nmx_c2 b) {
  nmx_c2 r;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a.y - a.x) in both halves, one instruction
NMX_DEV nmx_c2 nmx_pk_diff(nmx_c2 a) {
  nmx_c2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}

// window -> registers, NO cleaning (w, c wave-uniform).  Two halves so that a persistent kernel can issue the 16-byte
// loads of its NEXT item early (the part that goes to HBM) and fetch the successors / row ends -- bytes of the same
// cache lines -- when it starts on the item.
template <int WC = 0>
NMX_DEV __amdgpu_buffer_rsrc_t nmx_td_rsrc(const NmxTimeOscArgs& A, int w, int c) {
  // the window's start through the CONSTANT address space: an s_load (lgkmcnt) instead of a vector load whose
  // s_waitcnt vmcnt(0) would also wait for every load in flight (the table is written by the host before the launch)
  typedef const long long __attribute__((address_space(4)))* nmx_cll_p;
  const long long st = A.starts ? ((nmx_cll_p)(unsigned long long)A.starts)[w] : 0ll;
  const float* src = A.x + (long long)c * A.ch_stride + (long long)w * A.win_stride + st;
  return __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4 * (WC ? WC : A.W), 0x00020000);
}
template <int WC = 0>
NMX_DEV void nmx_td_load_x(const NmxTimeOscArgs& A, int w, int c, NmxTdRegs& R) {
  const int lane = (int)(threadIdx.x & 63);
  const __amdgpu_buffer_rsrc_t rs = nmx_td_rsrc<WC>(A, w, c);
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane + 1024 * k, 0, 0);
    R.x[k] = nmx_f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
  }
}
template <int WC = 0>
NMX_DEV void nmx_td_load_rest(const NmxTimeOscArgs& A, int w, int c, NmxTdRegs& R) {
  const int lane = (int)(threadIdx.x & 63);
  const __amdgpu_buffer_rsrc_t rs = nmx_td_rsrc<WC>(A, w, c);
#pragma unroll
  for (int k = 0; k < 4; ++k) R.s[k] = nmx_td_ldc2(rs, 16 * lane + 1024 * k + 16);
  R.first = nmx_td_ldc2(rs, 0);
  R.last = nmx_td_ldc2(rs, 4 * ((WC ? WC : A.W) - 2));
}
// The same successors / row ends WITHOUT memory: from the x registers of the neighbouring lanes (wave_shl:1; lane 63
// takes lane 0 of the next group).  For the persistent kernel: its item then starts with no vector-memory wait at all --
// loads retire in order, so a load issued at the top of an item also waits for the previous item's result stores.
template <int WC>
NMX_DEV void nmx_td_rest_from_x(NmxTdRegs& R) {
  static_assert(WC % 4 == 0 && WC >= 8 && WC <= 1024, "window length");
  const int lane = (int)(threadIdx.x & 63);
  auto shl1 = [](float v) {   // lane l <- lane l + 1, lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
  };
  auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  const bool top = lane == 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float nx = shl1(R.x[k][0]), ny = shl1(R.x[k][1]);
    if (k < 3) {
      const float wx = rl(R.x[k + 1][0], 0), wy = rl(R.x[k + 1][1], 0);
      nx = top ? wx : nx;
      ny = top ? wy : ny;
    }
    R.s[k] = nmx_mk2(nx, ny);
  }
  R.first = nmx_mk2(rl(R.x[0][0], 0), rl(R.x[0][1], 0));
  constexpr int g = WC / 4 - 1;   // the last group of four
  R.last = nmx_mk2(rl(R.x[g / 64][2], g % 64), rl(R.x[g / 64][3], g % 64));
}
template <int WC = 0>
NMX_DEV void nmx_td_load(const NmxTimeOscArgs& A, int w, int c, NmxTdRegs& R) {
  nmx_td_load_x<WC>(A, w, c, R);
  nmx_td_load_rest<WC>(A, w, c, R);
}

// 0 / 1 factors of a ragged group (W % 4 == 0, n0 % 4 == 0: a group of four is inside or outside the window as a
// whole): mx = the group is inside, m1 = so is the next one.  Of x0..x3, d1[0..3] and d2[0..3] only d1[3], d2[2] and
// d2[3] reach into the next group.
struct NmxTdMask {
  nmx_c2 xx, x1, n1;   // (mx, mx), (mx, m1), (m1, m1)
  NMX_DEV NmxTdMask(int n0, int W) {
    const float mx = n0 < W ? 1.f : 0.f, m1 = n0 + 4 < W ? 1.f : 0.f;
    xx = nmx_mk2(mx, mx); x1 = nmx_mk2(mx, m1); n1 = nmx_mk2(m1, m1);
  }
};

// Statistics of the window in R.  Returns false -- nothing written -- when the window holds a NaN or an infinity (the
// caller then runs the cleaning path).  `cen`: when not null, the centred window x - mean is written there (LDS,
// natural order, W floats; entries beyond W are not touched).
// WC: the window length when it is a compile-time constant (the reciprocals below then are constants too), 0 = A.W.
// SUM4: the four wave sums of pass 2 through nmx_wave_sum4 (14 instead of 32 instructions; off in the prefetching
// kernel, where the longer live range of the line-length sum costs it a register spill -- and a spill reload is a
// vector-memory load that waits for the prefetch).
// FEATC: the enabled features when the kernel is compiled for ONE set (0: read A.features) -- every feature test then
// folds away; a test that stays costs a scalar load, a compare and a branch of the wave's issue slots per item.
template <int WC = 0, bool SUM4 = true, unsigned FEATC = 0>
NMX_DEV bool nmx_td_emit(const NmxTimeOscArgs& A, int w, int c, NmxTdRegs& R, float* cen) {
  const unsigned features = FEATC ? FEATC : A.features;
  const int lane = (int)(threadIdx.x & 63);
  const int W = WC ? WC : A.W;
  auto add = [](float a, float b) { return a + b; };
  // ---- pass 1: sum x, first differences, sum |d1| ----------------------------------------------------------
  nmx_c2 D[4][2];
  nmx_c2 a0 = nmx_mk2(0.f, 0.f);
  float p3 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (256 * k >= W) {
      D[k][0] = D[k][1] = nmx_mk2(0.f, 0.f);
      continue;
    }
    const nmx_c2 X0 = R.x[k].xy, X1 = R.x[k].zw;
    const nmx_c2 Y0 = nmx_pk_hilo(X0, X1), Y1 = nmx_pk_hilo(X1, R.s[k]);   // (x1, x2), (x3, x4)
    a0 += X0;
    a0 += X1;
    D[k][0] = Y0 - X0;
    D[k][1] = Y1 - X1;
    if (!(4 * (63 + 64 * k) + 5 < W)) {   // ragged group: d1 of the samples without a successor := 0
      const NmxTdMask m(4 * (lane + 64 * k), W);
      D[k][0] *= m.xx;
      D[k][1] *= m.x1;
    }
    p3 += fabsf(D[k][0].x);
    p3 += fabsf(D[k][0].y);
    p3 += fabsf(D[k][1].x);
    p3 += fabsf(D[k][1].y);
  }
  const float p0 = nmx_wave_reduce(a0.x + a0.y, 0.f, add);
  // NaN / +-inf in the window, or samples on the rail (a cleaned infinity behind a re-reference: +-1e38s, whose squares are
  // not float32s): the scalar formulation with its two-pass forms (wave-uniform)
  if (!(fabsf(p0) < 1e30f)) return false;
  if (!SUM4) p3 = nmx_wave_reduce(p3, 0.f, add);
  R.sum = p0;
  const float rW = 1.f / (float)W, rW1 = 1.f / (float)(W - 1), rW2 = 1.f / (float)(W - 2);   // (scalar unit: W is uniform)
  const float m0 = p0 * rW;
  float* out_row = A.out + (long long)w * A.n_outputs;
  const bool hj = (features & NMXD_F_HJORTH) != 0;
  bool p3_done = !SUM4;
  if (hj || cen) {
    // sums of d1 and d2 telescope
    const float d_first = R.first.y - R.first.x, d_last = R.last.y - R.last.x;
    const float m1 = (R.last.y - R.first.x) * rW1, m2 = (d_last - d_first) * rW2;
    const nmx_c2 M0 = nmx_mk2(m0, m0), M1 = nmx_mk2(m1, m1), M2 = nmx_mk2(m2, m2);
    nmx_c2 q0 = nmx_mk2(0.f, 0.f), q1 = q0, q2 = q0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (256 * k >= W) continue;
      const bool full = 4 * (63 + 64 * k) + 5 < W;
      const int n0 = 4 * (lane + 64 * k);
      const nmx_c2 X0 = R.x[k].xy, X1 = R.x[k].zw;
      nmx_c2 E0 = X0 - M0, E1 = X1 - M0;
      const NmxTdMask m(n0, W);
      if (!full) { E0 *= m.xx; E1 *= m.xx; }
      if (cen && n0 < W) ((nmx_f4*)cen)[lane + 64 * k] = nmx_f4{E0.x, E0.y, E1.x, E1.y};
      if (!hj) continue;
      q0 = nmx_c2_fma(E0, E0, q0);
      q0 = nmx_c2_fma(E1, E1, q0);
      nmx_c2 F0 = D[k][0] - M1, F1 = D[k][1] - M1;
      // second differences = differences of the first ones, d1[4] = x5 - x4 (the shifted pairs are rebuilt here
      // rather than kept from pass 1: registers)
      const nmx_c2 S0 = nmx_pk_hilo(D[k][0], D[k][1]), S1 = nmx_pk_hilo(D[k][1], nmx_pk_diff(R.s[k]));
      nmx_c2 G0 = S0 - D[k][0] - M2, G1 = S1 - D[k][1] - M2;
      if (!full) {
        F0 *= m.xx; F1 *= m.x1;
        G0 *= m.xx; G1 *= m.n1;
      }
      q1 = nmx_c2_fma(F0, F0, q1);
      q1 = nmx_c2_fma(F1, F1, q1);
      q2 = nmx_c2_fma(G0, G0, q2);
      q2 = nmx_c2_fma(G1, G1, q2);
    }
    if (hj) {
      float s0 = q0.x + q0.y, s1 = q1.x + q1.y, s2 = q2.x + q2.y;
      if (SUM4) {
        nmx_wave_sum4(p3, s0, s1, s2);   // (the line length rides along)
        p3_done = true;
      } else {
        s0 = nmx_wave_reduce(s0, 0.f, add); s1 = nmx_wave_reduce(s1, 0.f, add); s2 = nmx_wave_reduce(s2, 0.f, add);
      }
      if (lane == 0) {
        const float v0 = s0 * rW, v1 = s1 * rW1, v2 = s2 * rW2;
        float act, mob, comp;
        const bool normal = v0 > 1e-30f && v0 < 1e30f && v1 > 1e-30f && v1 < 1e30f && v2 < 1e30f;   // (wave-uniform)
        if (normal) {
          act = v0;
          const float r1 = v1 * __builtin_amdgcn_rcpf(v0), r2 = v2 * __builtin_amdgcn_rcpf(v1);
          mob = __builtin_amdgcn_sqrtf(r1);
          comp = __builtin_amdgcn_sqrtf(r2) * __builtin_amdgcn_rcpf(mob);
        } else {   // flat / degenerate windows: the reference's nan_to_num placement on IEEE arithmetic
          act = nmx_nan_to_num(v0);
          mob = nmx_nan_to_num(sqrtf(v1 / v0));
          comp = nmx_nan_to_num(sqrtf(v2 / v1) / mob);
        }
        const int col = A.hjorth_cols.base + c * A.hjorth_cols.ch_stride;
        out_row[col] = act;
        out_row[col + A.hjorth_cols.a_stride] = mob;
        out_row[col + 2 * A.hjorth_cols.a_stride] = comp;
      }
    }
  }
  if (!p3_done && (features & NMXD_F_LINELENGTH)) p3 = nmx_wave_reduce(p3, 0.f, add);
  if (lane == 0) {
    if (features & NMXD_F_LINELENGTH) out_row[A.ll_cols.base + c * A.ll_cols.ch_stride] = p3 * rW1 * rW1;
    if (features & NMXD_F_RAW) out_row[A.raw_cols.base + c * A.raw_cols.ch_stride] = R.last.y + nmx_dc_of(A, c);
  }
  return true;
}

// one (window, channel) of the scan kernel (no oscillatory feature enabled)
NMX_DEV void nmx_td_item(const NmxTimeOscArgs& A, int w, int c) {
  w = nmx_uniform_i(w);
  c = nmx_uniform_i(c);
  if (nmx_td_ok(A)) {
    NmxTdRegs Rt;
    nmx_td_load(A, w, c, Rt);
    if (nmx_td_emit(A, w, c, Rt, nullptr)) return;
  }
  NmxScanRegs R;   // odd window lengths, or a NaN / infinity in the window
  nmx_scan_load(A, w, c, R);
  nmx_scan_emit(A, w, c, R);
}
#endif

// nmx_k_burst_stat_reg.h -- kernel D of the bursts chain (nmx_k_bursts.h: nmx_burst_stat_item) with the envelope of one
// (window, channel, band) in the REGISTERS of one wave (device only): W % 4 == 0, W <= 64 * CH.
//
// Reference: features/bursts.py:171-258 (threshold -> runs -> six statistics), same run walk as nmx_burst_stat_item:
// lane l owns the CH consecutive samples [l CH, l CH + CH) (CH / 4 16-byte loads straight from the envelope buffer:
// no LDS staging, no padded LDS reads), pass 1 = chunk sum + last below-threshold position, DPP scans carry the
// latest zero and the open run's sum across lanes, pass 2 = the run walk, both passes unrolled over the chunk.  Run sums
// in fp32 (short chains: see the scan below); the LDS form keeps the double prefix sums.
#pragma once

#include "nmx_k_bank_w64.h"   // nmx_rsrc
#include "nmx_k_bursts.h"

#ifndef NMX_HOST_EMU
template <int CH>
NMX_DEV void nmx_burst_stat_item_reg(const NmxBurstStatArgs& A, int w, int c, int bi) {
  const int l = NMX_TID, W = A.W;
  const long long item = ((long long)w * A.n_channels + c) * A.n_bands + bi;
  const nmx_rsrc rs = nmx_make_rsrc(A.env + item * W, 4 * W);   // reads past the window return 0
  float v[CH];
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, 4 * (l * CH + 4 * q), 0, 0);
    v[4 * q] = __uint_as_float(t.x); v[4 * q + 1] = __uint_as_float(t.y);
    v[4 * q + 2] = __uint_as_float(t.z); v[4 * q + 3] = __uint_as_float(t.w);
  }
  const float thr = A.thr[item];
  const int i0 = l * CH;
  const int n = W - i0 < 0 ? 0 : (W - i0 > CH ? CH : W - i0);   // live samples of this lane
  // pass 1: last below-threshold position in the chunk and the sum of the samples after it (the open run's part here)
  float tail = 0.f;
  int zpos = -1;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const bool live = j < n;
    const bool z = live && !(v[j] >= thr);
    tail = z ? 0.f : (live ? tail + v[j] : tail);
    zpos = z ? i0 + j : zpos;
  }
  // segmented scan over the lanes (DPP, Hillis-Steele in the rows, row_bcast across them): a block of lanes is
  // (latest zero in it or -1, sum of its samples after that zero); a block on the right without a zero continues the
  // left one.  Run sums are thus formed from at most CH sequential + 6 tree additions in fp32 (the prefix-difference
  // form of nmx_burst_stat_item needs doubles: sixteen v_add_f64 + conversions + 64-bit selects per lane and pass).
  int lastz = zpos;
  float open = tail;
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    const int tp = nmx_dpp_i(lastz, -1, st);
    const float ts = __builtin_bit_cast(float, nmx_dpp_i(__builtin_bit_cast(int, open), 0, st));
    open = lastz >= 0 ? open : open + ts;
    lastz = lastz >= 0 ? lastz : tp;
  }
  open = __builtin_bit_cast(float, nmx_dpp_i(__builtin_bit_cast(int, open), 0, 6));   // exclusive: wave_shr:1
  lastz = nmx_dpp_i(lastz, -1, 6);
  // the sample before my chunk: the last one of the previous lane (every earlier lane is full when I have samples)
  const int pb = __builtin_amdgcn_update_dpp(0, (int)(v[CH - 1] >= thr), 0x138, 0xf, 0xf, false);   // wave_shr:1
  bool prev = l > 0 && pb != 0;
  // pass 2: the run walk
  int n_above = 0, n_trans = 0, n_valid = 0, max_len = 0;
  float sum_means = 0.f, run = open;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    if (j < n) {
      const float x = v[j];
      const bool b = x >= thr;
      n_trans += (b != prev) ? 1 : 0;
      n_above += b ? 1 : 0;
      amax = (b && x > amax) ? x : amax;
      if (!b && prev) {   // the run [lastz + 1, i - 1] just finished -> a valid run
        const int len = i0 + j - 1 - lastz;
        sum_means += run * __builtin_amdgcn_rcpf((float)len);   // (v_rcp_f32, 1 ulp: the mean is an fp32 output)
        ++n_valid;
        max_len = len > max_len ? len : max_len;
      }
      lastz = b ? lastz : i0 + j;
      run = b ? run + x : 0.f;
      prev = b;
    }
  }
  float fa = (float)n_above, ft = (float)n_trans, fv = (float)n_valid;   // exact below 2^24
  float fh = sum_means;
  nmx_wave_sum4(fa, ft, fv, fh);
  const int mlen = nmx_wave_reduce(max_len, 0, [](int a, int b) { return a > b ? a : b; });
  amax = nmx_wave_reduce(amax, 0.f, [](float a, float b) { return a > b ? a : b; });
  const int last_lane = (W - 1) / CH;   // wave-uniform
  const int in_burst = __builtin_amdgcn_readlane((int)prev, __builtin_amdgcn_readfirstlane(last_lane));
  if (l == 0) {
    const int num_bursts = (int)ft / 2, nv = (int)fv;
    const float dmean = num_bursts ? (fa / (float)num_bursts) / A.sfreq : 0.f;
    float vals[6];
    vals[0] = dmean;
    vals[1] = (float)mlen / A.sfreq;
    vals[2] = nv ? fh / (float)nv : 0.f;
    vals[3] = amax;
    vals[4] = dmean / A.seg_s;
    vals[5] = in_burst ? 1.f : 0.f;
    float* row = A.out + (long long)w * A.n_outputs;
    int col = A.cols.base + c * A.cols.ch_stride + bi * A.cols.a_stride;
#pragma unroll
    for (int s = 0; s < 6; ++s)
      if (A.out_mask & (1u << s)) {
        row[col] = vals[s];
        col += A.cols.b_stride;
      }
  }
}
#endif

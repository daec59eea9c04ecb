// nmx_emu.cpp -- TEST-ONLY single-thread logic emulator of the nmx kernels.
//
// Compiles the SAME device source (py_neuromodulation_amd/csrc/nmx_k_*.h) with
// -DNMX_HOST_EMU (one "thread" per workgroup, barriers are no-ops) behind the same C ABI,
// so the CPU-only test suite can check kernel *logic* (index math, FFT staging, epilogues)
// against the oracle in a container without a GPU.  It is NOT part of the product: the
// package loader only ever opens libnmx.so (HIP); nothing under py_neuromodulation_amd/
// references this file, and wave-level behaviour (races, barriers, shuffles) is only
// exercised by the -m gpu tests on the MI355X.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define NMX_HOST_EMU 1
#include "../../py_neuromodulation_amd/csrc/nmx_k_bank.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_bank_w64.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_bursts.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_burst_fill.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_kalman.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_norm.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_power.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_prep.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_rawnorm.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_resample.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_resample64.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_sharpwave.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_timeosc.h"
#include "../../py_neuromodulation_amd/csrc/nmx_k_specmm.h"

typedef void* be_stream_t;
struct be_timer_t { bool used = false; };
static int nmx_fail(int code, const std::string& msg);

static int be_device_count() { return 1; }
static int be_set_device(int) { return 0; }
static void* be_alloc(size_t n) { return calloc(1, n ? n : 4); }
static void be_free(void* p) { free(p); }
static void be_dev_pool_age() {}
static long long be_dev_pool_trim(long long) { return 0; }
static void* be_host_alloc(size_t n) { return malloc(n ? n : 4); }
static void be_host_free(void* p) { free(p); }
static void be_h2d_sync(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void be_d2h_sync(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void be_memset_sync(void* d, int v, size_t n) { memset(d, v, n); }
static void be_h2d_async(void* d, const void* s, size_t n, be_stream_t) { memcpy(d, s, n); }
static void be_d2h_async(void* d, const void* s, size_t n, be_stream_t) { memcpy(d, s, n); }
static void be_h2d_2d_async(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, be_stream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
}
static void be_d2h_2d_async(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, be_stream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
}
static void be_memset_async(void* d, int v, size_t n, be_stream_t) { memset(d, v, n); }
static void be_host_fn(be_stream_t, void (*fn)(void*), void* arg) { fn(arg); }
static int be_sync(be_stream_t) { return 0; }
static void be_sync_quiet(be_stream_t) {}
static int be_sync_watch(be_stream_t, const std::string*, int) { return 0; }
static be_stream_t be_stream_create() { return nullptr; }
static be_stream_t be_stream_create_high() { return nullptr; }
static void be_stream_destroy(be_stream_t) {}
typedef int be_event_t;
static void be_event_create(be_event_t&) {}
static void be_event_destroy(be_event_t&) {}
static void be_event_record(be_event_t&, be_stream_t) {}
static void be_stream_wait(be_stream_t, be_event_t&) {}
static void be_timer_create(be_timer_t&) {}
static void be_timer_destroy(be_timer_t&) {}
static void be_timer_start(be_timer_t&, be_stream_t) {}
static void be_timer_stop(be_timer_t& t, be_stream_t) { t.used = true; }
static float be_timer_elapsed(be_timer_t&) { return 0.f; }
static int be_check_launch() { return 0; }
static void be_stage(int) {}
static void be_stage_reset() {}
static std::string be_stage_kernels(int) { return "host emulator (tests only)"; }

// the matrix-pipe spectrum kernel's arithmetic and flag protocol (nmx_k_specmm.h): a 16-bit mask per tile of 16 windows of one
// channel (tile = group * n_channels + channel), the flagged windows redone by the generic item code with its cleaning
static int be_launch_timeosc(const NmxTimeOscArgs& A, int n_items, int, size_t lds, be_stream_t) {
  static const bool smm = [] { const char* v = getenv("NMX_SPECMM"); return !(v && v[0] == '0'); }();
  if (smm && A.smm_tab && nmx_specmm_ok(A)) {
    const int C = A.n_channels, n_windows = n_items / C;
    for (int t = 0; t < ((n_windows + 15) / 16) * C; ++t) A.todo[t] = 0;
    for (int w = 0; w < n_windows; ++w)
      for (int c = 0; c < C; ++c)
        if (nmx_specmm_item_emu(A, w, c)) A.todo[(w / 16) * C + c] |= (unsigned short)(1u << (w & 15));
    return 1;
  }
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) nmx_time_osc_item(A, it / A.n_channels, it % A.n_channels, sm.data());
  return 0;
}
static void be_launch_timeosc_redo(const NmxTimeOscArgs& A, int n_items, be_stream_t) {
  const int C = A.n_channels, n_windows = n_items / C;
  std::vector<float> sm((size_t)A.lds_floats + 16);
  for (int w = 0; w < n_windows; ++w)
    for (int c = 0; c < C; ++c)
      if (A.todo[(w / 16) * C + c] & (1u << (w & 15))) nmx_time_osc_item(A, w, c, sm.data());
}
static void be_launch_bank(const NmxBankArgs& A, int n_items, int, size_t lds, be_stream_t) {
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) nmx_bank_item(A, it / A.n_channels, it % A.n_channels, sm.data());
}
static void be_launch_bank_w64(const NmxBankW64Args& A, int n_items, size_t lds, be_stream_t) {
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) {
    if (A.b.pad_mode == 0) nmx_bank_w64_item<0, 0, 1>(A, it / A.b.n_channels, it % A.b.n_channels, sm.data(), nullptr);
    else nmx_bank_w64_item<1, 0, 0>(A, it / A.b.n_channels, it % A.b.n_channels, sm.data(), nullptr);
  }
}
static bool be_bank_w64_takes_dc(const NmxBankW64Args&, int) { return false; }   // (the one-wave item code reads a copy)
static bool be_timeosc_takes_dc(const NmxTimeOscArgs&) { return true; }
static void be_launch_sharp_todo(const NmxSharpArgs&, int, size_t, const unsigned char*, be_stream_t) {}
static void be_launch_sharp_dense(const NmxSharpArgs&, int, be_stream_t) {}
static void be_launch_hilbert(const NmxHilbertArgs& A, long long n_items, int, size_t lds, be_stream_t) {
  std::vector<float> sm(lds / 4 + 16);
  for (long long it = 0; it < n_items; ++it) nmx_hilbert_item(A, it, sm.data());
}
static void be_launch_burst_thr(const NmxBurstThrArgs& A, int n_items, int, size_t lds, be_stream_t, long long = -1) {
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) nmx_burst_thr_item<1>(A, it / A.n_bands, it % A.n_bands, sm.data());
}
static void be_launch_burst_fill(const NmxBurstThrArgs& A, int n_items, unsigned short*, float*, be_stream_t) {
  for (int it = 0; it < n_items; ++it) nmx_burst_fill_item_emu(A, it / A.n_bands, it % A.n_bands);
}
static void be_launch_burst_stat(const NmxBurstStatArgs& A, int n_items, size_t lds, be_stream_t) {
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) {
    const int bi = it % A.n_bands, r = it / A.n_bands;
    nmx_burst_stat_item(A, r / A.n_channels, r % A.n_channels, bi, sm.data());
  }
}
static void be_launch_sharp(const NmxSharpArgs& A, int n_items, size_t lds, be_stream_t) {
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) {
    const int fi = it % A.n_filters, r = it / A.n_filters;
    nmx_sharp_item(A, r / A.n_channels, r % A.n_channels, fi, sm.data());
  }
}
static void be_launch_reref(const NmxRerefArgs& A, be_stream_t) {
  for (int c0 = 0; c0 < A.C; c0 += NMX_REREF_ROWS)
    for (long long t = 0; t < A.T; ++t) nmx_reref_tile(A, t, c0);
}
static void be_launch_car(const NmxCarArgs& A, be_stream_t) {
  for (long long t = 0; t < A.T; ++t) nmx_car_sample(A, t);
}
static void be_launch_reref_struct(const NmxRerefStructArgs& A, be_stream_t) {
  for (long long t = 0; t < A.T; ++t) nmx_reref_struct_sample(A, t);
}
static void be_launch_resample(const NmxResampleArgs& A, int n_items, int, size_t lds, be_stream_t) {
  std::vector<float> sm(lds / 4 + 16);
  for (int it = 0; it < n_items; ++it) nmx_resample_item(A, it / A.n_channels, it % A.n_channels, sm.data());
}
static void be_launch_rawnorm(const NmxRawNormArgs& A, be_stream_t) {
  if (A.method >= NMX_RAWNORM_MEDIAN && A.method != NMX_RAWNORM_POWER) {
    std::vector<float> sm(6 * (size_t)A.max_list + 4 * NMX_RAWNORM_ORDER_NT + 288);
    for (int c = 0; c < A.n_channels; ++c) nmx_rawnorm_order_item(A, c, sm.data());
  } else
  for (int c = 0; c < A.n_channels; ++c) nmx_rawnorm_stats_item(A, c);
  const long long n = (long long)A.n_windows * A.n_channels * A.W;
  for (long long i = 0; i < n; ++i) nmx_rawnorm_apply(A, i);
}
static void be_launch_kalman(const NmxKalmanArgs& A, be_stream_t) {
  for (int c = 0; c < A.n_channels; ++c)
    for (int b = 0; b < A.n_bands; ++b) nmx_kalman_item(A, c, b);
}
static void be_launch_norm(const NmxNormArgs& A, be_stream_t) {
  for (int j = 0; j < A.n_cols; ++j) nmx_norm_column(A, j);
}
static void be_launch_norm_scan(const NmxNormArgs& A, const NmxNormScan& S, be_stream_t) {
  for (int g = 0; g < S.n_hseg; ++g)
    for (int j = 0; j < A.n_cols; ++j) nmx_norm_seg_hist(A, S, g, j);
  for (int g = 0; g < S.n_bseg; ++g)
    for (int j = 0; j < A.n_cols; ++j) nmx_norm_seg_batch(A, S, g, j);
  for (int j = 0; j < A.n_cols; ++j) nmx_norm_seg_offsets(A, S, j);
  for (int r = 0; r < A.n_rows; ++r)
    for (int j = 0; j < A.n_cols; ++j) nmx_norm_scan_cell(A, S, r, j);
  for (int r = 0; r < A.n_rows; ++r)
    for (int j = 0; j < A.n_cols; ++j) nmx_norm_scan_ring(A, S, r, j);
}
static void be_launch_power(const NmxPowerPrepArgs& P, const NmxPowerArgs& A, be_stream_t) {
  for (int e = 0; e < P.have + P.n_rows; ++e)
    for (int j = 0; j < P.n_cols; ++j) nmx_power_prep_at(P, e, j);
  for (int r = 0; r < A.n_rows; ++r)
    for (int j = 0; j < A.n_cols; ++j) nmx_power_cell(A, r, j);
  for (int r = 0; r < P.n_rows; ++r)
    for (int j = 0; j < P.n_cols; ++j) nmx_power_ring_at(P, r, j);
}
static void be_launch_shift(const NmxShiftArgs& A, be_stream_t) {
  for (int c = 0; c < A.C; ++c)
    for (long long t = 0; t < A.T; ++t) nmx_shift_sample(A, t, c);
}
static void be_launch_reref64(const NmxReref64Args& A, be_stream_t) {
  for (int c0 = 0; c0 < A.C; c0 += NMX_REREF64_ROWS)
    for (long long t = 0; t < A.T; ++t) nmx_reref64_tile(A, t, c0);
}
static void be_launch_rs64_elem(const NmxResample64Args& A, int mode, const NmxCplx64* src, NmxCplx64* dst, long long n, int rows,
                                be_stream_t) {
  for (int c = 0; c < rows; ++c)
    for (long long i = 0; i < n; ++i) nmx_rs64_elem(A, mode, src, dst, c, i);
}
static void be_launch_rs64_pass(const NmxCplx64* src, NmxCplx64* dst, long long ld, long long n, long long ns, long long st, int sign,
                                int rows, be_stream_t) {
  for (int c = 0; c < rows; ++c)
    for (long long t = 0; t < n / 2; ++t) nmx_rs64_pass(src, dst, ld, n, ns, st, sign, c, t);
}
static void be_launch_nanmask(const NmxNanMaskArgs& A, int n_items, be_stream_t) {
  float sm[64];
  for (int it = 0; it < n_items; ++it) nmx_nanmask_item(A, it / A.C_in, it % A.C_in, sm);
}
static void be_launch_tap(const NmxTapArgs& A, int n_items, be_stream_t) {
  for (int it = 0; it < n_items; ++it) nmx_tap_item(A, it / A.C, it % A.C);
}

#include "../../py_neuromodulation_amd/csrc/nmx_engine.inc"

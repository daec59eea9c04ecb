// nmx_api.hip -- libnmx.so: HIP (gfx950) backend + C ABI.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC nmx_api.hip -o ../libnmx.so
// There is no CPU path in this library: every entry point that computes launches kernels.
#include <chrono>
#include <map>
#include <unordered_map>
#include <thread>
#include <hip/hip_runtime.h>

#include "nmx_k_bank.h"
#include "nmx_k_bank_w64.h"
#include "nmx_k_bursts.h"
#include "nmx_k_burst_fill.h"
#include "nmx_k_kalman.h"
#include "nmx_k_norm.h"
#include "nmx_k_power.h"
#include "nmx_k_prep.h"
#include "nmx_k_rawnorm.h"
#include "nmx_k_resample.h"
#include "nmx_k_resample64.h"
#include "nmx_k_sharpwave.h"
#include "nmx_k_timeosc.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

// ---- kernels: one workgroup per item, dynamic LDS carved by the host plan -----------------
extern __shared__ __attribute__((aligned(16))) float nmx_smem[];

__global__ void __launch_bounds__(256) nmx_kern_timeosc(const NmxTimeOscArgs A) {
  const int item = blockIdx.x;
  nmx_time_osc_item(A, item / A.n_channels, item % A.n_channels, nmx_smem);
}
__global__ void __launch_bounds__(256) nmx_kern_bank(const NmxBankArgs A, int n_items) {
  // one item per workgroup; partitioned mode (NmxBankArgs::ups_*): a workgroup walks items with its scratch slot
  for (int item = blockIdx.x; item < n_items; item += gridDim.x)
    nmx_bank_item(A, item / A.n_channels, item % A.n_channels, nmx_smem, (int)blockIdx.x);
}
__global__ void __launch_bounds__(256) nmx_kern_hilbert(const NmxHilbertArgs A) {
  nmx_hilbert_item(A, (long long)blockIdx.x, nmx_smem);
}
template <int CH>
__global__ void __launch_bounds__(256) nmx_kern_burst_thr(const NmxBurstThrArgs A) {
  // the sequential threshold walk sits on the critical path of the overlapped schedule: its few waves
  // win issue arbitration against the throughput kernels that share the CU
  __builtin_amdgcn_s_setprio(3);
  const int item = blockIdx.x;
  nmx_burst_thr_item<CH>(A, item / A.n_bands, item % A.n_bands, nmx_smem);
}
// fresh stream: the fill phase of the history as one sort + a barrier-free walk (nmx_k_burst_fill.h)
__global__ void __launch_bounds__(NMX_FILL_NT) nmx_kern_burst_fill(const NmxBurstThrArgs A, int n2, unsigned short* slots) {
  const int item = blockIdx.x;
  nmx_burst_fill_item(A, item / A.n_bands, item % A.n_bands, n2, slots, nmx_smem);
}
__global__ void __launch_bounds__(NMX_FILL_NT) nmx_kern_burst_fill_sort(const NmxBurstThrArgs A, int n2, unsigned short* slots, float* sorted) {
  const int item = blockIdx.x;
  nmx_burst_fill_sort_item(A, item / A.n_bands, item % A.n_bands, n2, slots, sorted, nmx_smem);
}
__global__ void __launch_bounds__(64) nmx_kern_burst_fill_walk(const NmxBurstThrArgs A, int n2, const unsigned short* slots, const float* sorted) {
  const int item = blockIdx.x;
  nmx_burst_fill_walk_item(A, item / A.n_bands, item % A.n_bands, n2, slots, sorted, nmx_smem);
}
// long histories (4 kHz x 30 s at the 60th percentile: 48 001 list entries): 1024 threads x 64 entries each
__global__ void __launch_bounds__(1024) nmx_kern_burst_thr_wide(const NmxBurstThrArgs A) {
  const int item = blockIdx.x;
  nmx_burst_thr_item<64>(A, item / A.n_bands, item % A.n_bands, nmx_smem);
}
// kernels compiled with a compile-time workgroup size live in nmx_timeosc.hip / nmx_wave.hip
extern "C" void nmx_timeosc_fixed_launch128(const NmxTimeOscArgs* A, int n_items, size_t lds, hipStream_t s);
extern "C" void nmx_hilbert_fixed_launch128(const NmxHilbertArgs* A, long long n_items, size_t lds, hipStream_t s);
// one-item-per-wave kernels live in nmx_wave.hip (compile-time workgroup size)
extern "C" void nmx_wave_launch_burst_stat(const NmxBurstStatArgs* A, int n_items, size_t lds, hipStream_t s);
extern "C" void nmx_wave_launch_sharp(const NmxSharpArgs* A, int n_items, size_t lds, hipStream_t s);
__global__ void __launch_bounds__(256) nmx_kern_reref(const NmxRerefArgs A) {
  nmx_reref_tile(A, (long long)blockIdx.x * 256 + threadIdx.x, (int)blockIdx.y * NMX_REREF_ROWS);
}
__global__ void __launch_bounds__(256) nmx_kern_resample(const NmxResampleArgs A) {
  const int item = blockIdx.x;
  nmx_resample_item(A, item / A.n_channels, item % A.n_channels, nmx_smem);
}
__global__ void __launch_bounds__(64) nmx_kern_rawnorm_stats(const NmxRawNormArgs A) {
  nmx_rawnorm_stats_item(A, (int)blockIdx.x);
}
__global__ void __launch_bounds__(NMX_RAWNORM_ORDER_NT) nmx_kern_rawnorm_order(const NmxRawNormArgs A) {
  extern __shared__ __attribute__((aligned(16))) float nmx_smem_rn[];
  nmx_rawnorm_order_item(A, (int)blockIdx.x, nmx_smem_rn);
}
__global__ void __launch_bounds__(256) nmx_kern_rawnorm_apply(const NmxRawNormArgs A) {
  nmx_rawnorm_apply(A, (long long)blockIdx.x * 256 + threadIdx.x);
}
__global__ void __launch_bounds__(64) nmx_kern_kalman(const NmxKalmanArgs A) {
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  nmx_kalman_item(A, i / A.n_bands, i % A.n_bands);
}
__global__ void __launch_bounds__(64) nmx_kern_norm(const NmxNormArgs A) {
  nmx_norm_column(A, (int)(blockIdx.x * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(64) nmx_kern_norm_seg_hist(const NmxNormArgs A, const NmxNormScan S) {
  nmx_norm_seg_hist(A, S, (int)blockIdx.y, (int)(blockIdx.x * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(64) nmx_kern_norm_seg_batch(const NmxNormArgs A, const NmxNormScan S) {
  nmx_norm_seg_batch(A, S, (int)blockIdx.y, (int)(blockIdx.x * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(64) nmx_kern_norm_seg_offsets(const NmxNormArgs A, const NmxNormScan S) {
  nmx_norm_seg_offsets(A, S, (int)(blockIdx.x * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(256) nmx_kern_norm_cell(const NmxNormArgs A, const NmxNormScan S) {
  nmx_norm_scan_cell(A, S, (int)blockIdx.x, (int)(blockIdx.y * 256 + threadIdx.x));
}
__global__ void __launch_bounds__(256) nmx_kern_norm_ring(const NmxNormArgs A, const NmxNormScan S) {
  nmx_norm_scan_ring(A, S, (int)blockIdx.x, (int)(blockIdx.y * 256 + threadIdx.x));
}
// rows on grid.x (2^31 - 1 blocks), column blocks on grid.y: grid.y / .z stop at 65 535, and a long offline table or a
// long history has more rows than that
__global__ void __launch_bounds__(256) nmx_kern_power_prep(const NmxPowerPrepArgs P) {
  nmx_power_prep_at(P, (int)blockIdx.x, (int)(blockIdx.y * 256 + threadIdx.x));
}
__global__ void __launch_bounds__(64) nmx_kern_power(const NmxPowerArgs A) {
  nmx_power_cell(A, (int)blockIdx.x, (int)(blockIdx.y * 64 + threadIdx.x));
}
__global__ void __launch_bounds__(256) nmx_kern_power_ring(const NmxPowerPrepArgs P) {
  nmx_power_ring_at(P, (int)blockIdx.x, (int)(blockIdx.y * 256 + threadIdx.x));
}
template <int NW>
__global__ void __launch_bounds__(64 * NW) nmx_kern_car(const NmxCarArgs A) {
  __shared__ double red[64 * NW];
  nmx_car_tile<NW>(A, (long long)blockIdx.x * 64, red);
}
__global__ void __launch_bounds__(256) nmx_kern_reref_struct(const NmxRerefStructArgs A) {
  __shared__ double red[NMX_RS_GROUPS * 256];
  nmx_reref_struct_tile(A, (long long)blockIdx.x * 64, red);
}
__global__ void __launch_bounds__(256) nmx_kern_shift(const NmxShiftArgs A) {
  nmx_shift_sample(A, (long long)blockIdx.x * 256 + threadIdx.x, (int)blockIdx.y);
}
__global__ void __launch_bounds__(64) nmx_kern_nanmask(const NmxNanMaskArgs A) {
  const int item = blockIdx.x;
  nmx_nanmask_item(A, item / A.C_in, item % A.C_in, nmx_smem);
}
__global__ void __launch_bounds__(256) nmx_kern_tap(const NmxTapArgs A) {
  const int item = blockIdx.x;
  nmx_tap_item(A, item / A.C, item % A.C);
}

// ---- which kernels ran (per stage of the launch sequence; stage indices = nmx_last_timing_ms) ----
static thread_local int g_stage = 0;
static thread_local std::string g_stage_kernels[8];
extern "C" void nmxi_note_kernel(const char* name) {
  std::string& s = g_stage_kernels[g_stage & 7];
  if (s.find(name) != std::string::npos) return;
  if (!s.empty()) s += " + ";
  s += name;
}
static void be_stage(int st) { g_stage = st; }
static void be_stage_reset() { for (auto& s : g_stage_kernels) s.clear(); }
static std::string be_stage_kernels(int st) { return g_stage_kernels[st & 7]; }

// ---- backend ------------------------------------------------------------------------------
typedef hipStream_t be_stream_t;
struct be_timer_t {
  hipEvent_t a = nullptr, b = nullptr;
  bool used = false;
};

static thread_local std::string g_be_err;
static thread_local int g_be_rc = 0;   // per host thread: the plans of a multi-device stream run on one thread each
static int nmx_fail(int code, const std::string& msg);
static int be_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  return nmx_fail(NMX_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define BE_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess && !g_be_rc) g_be_rc = be_hip(e_, #call); } while (0)

static int be_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
static int be_set_device(int dev) { return be_hip(hipSetDevice(dev), "hipSetDevice"); }
// Device allocations are recycled across plans.  The reference builds a fresh DataProcessor per Stream and per run
// (stream/stream.py:130, 233-242): a plan's ~50 hipMalloc and its owner's ~35 hipFree (each an unmap and a device
// synchronisation, ~0.1 ms) were 10 of the 39 ms a fresh Stream on a warm process costs.  A freed block goes to a
// per-device list keyed by its exact size -- identical streams ask for identical sizes -- after the SAME device-wide
// synchronisation hipFree implies (the engine leans on it: a hand-off buffer regrown in mid-batch may still be read by
// kernels in flight), so an idle block has no reader on any stream and may be handed to the driver from any thread.
// What stays cached is bounded three ways (co-tenants of the device -- a second rank, torch's allocator -- cannot reclaim
// it themselves):
//   * NMX_DEVICE_POOL_MB (default 8192; 0: off) AND an eighth of what the device has free when a block comes back
//     (hipMemGetInfo: on a device somebody else has filled the pool shrinks to nothing);
//   * age: a block no plan has asked for while NMX_POOL_KEEP_PLANS (4) plans were destroyed goes back to the driver
//     (a test suite, a sweep over window lengths: sizes that never recur);
//   * nmx_device_pool_trim(keep_bytes): the caller's "give it back now" (engine.release_staging(), atexit).
// A failed hipMalloc empties the lists and tries again.  Blocks come back with their old contents, as hipMalloc's may.
struct NmxDevIdle { void* p; long long gen; };
struct NmxDevPool {
  std::mutex m;
  std::multimap<std::pair<int, size_t>, NmxDevIdle> idle;
  std::map<void*, std::pair<int, size_t>> live;
  size_t cached = 0, cap = 0;
  long long gen = 0;   // plans destroyed so far
  bool cap_read = false;
};
static NmxDevPool& be_dev_pool() { static NmxDevPool* p = new NmxDevPool(); return *p; }   // (leaked: outlives static destruction)
static void be_dev_pool_flush(NmxDevPool& D, size_t keep = 0, long long older_than = -1) {   // (caller holds the lock)
  for (auto it = D.idle.begin(); it != D.idle.end();) {
    const bool old = older_than >= 0 && it->second.gen <= older_than;
    if (!(old || (older_than < 0 && D.cached > keep))) { ++it; continue; }
    (void)hipFree(it->second.p);
    D.cached -= it->first.second;
    it = D.idle.erase(it);
  }
}
static void be_dev_pool_cap(NmxDevPool& D) {   // (caller holds the lock)
  if (D.cap_read) return;
  const char* v = getenv("NMX_DEVICE_POOL_MB");
  D.cap = (size_t)((v && v[0] >= '0' && v[0] <= '9') ? atoll(v) : 8192) << 20;
  D.cap_read = true;
}
static void* be_alloc(size_t n) {
  NmxDevPool& D = be_dev_pool();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const size_t want = ((n ? n : 4) + 255) & ~(size_t)255;
  std::lock_guard<std::mutex> lk(D.m);
  be_dev_pool_cap(D);
  auto it = D.idle.find(std::make_pair(dev, want));
  if (it != D.idle.end()) {
    void* p = it->second.p;
    D.idle.erase(it);
    D.cached -= want;
    D.live[p] = std::make_pair(dev, want);
    return p;
  }
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    be_dev_pool_flush(D);
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  }
  D.live[p] = std::make_pair(dev, want);
  return p;
}
static void be_free(void* p) {
  if (!p) return;
  NmxDevPool& D = be_dev_pool();
  (void)hipDeviceSynchronize();   // what hipFree did: nothing in flight reads the block any more
  std::lock_guard<std::mutex> lk(D.m);
  auto it = D.live.find(p);
  if (it == D.live.end()) { (void)hipFree(p); return; }
  const std::pair<int, size_t> key = it->second;
  D.live.erase(it);
  size_t cap = D.cap;
  if (cap && key.second >= (size_t)(64u << 20)) {   // (the query costs microseconds: asked for the blocks that matter)
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) cap = std::min(cap, (fr + D.cached) / 8);
    else (void)hipGetLastError();
  }
  if (cap && D.cached + key.second <= cap) {
    D.idle.emplace(key, NmxDevIdle{p, D.gen});
    D.cached += key.second;
    return;
  }
  (void)hipFree(p);
}
// a plan is gone: blocks that sat idle through the last `keep_plans` destroyed plans go back to the driver
static void be_dev_pool_age() {
  NmxDevPool& D = be_dev_pool();
  static const long long keep_plans = [] {
    const char* v = getenv("NMX_POOL_KEEP_PLANS");
    return (long long)((v && v[0] >= '0' && v[0] <= '9') ? atoll(v) : 4);
  }();
  std::lock_guard<std::mutex> lk(D.m);
  D.gen += 1;
  if (D.gen > keep_plans) be_dev_pool_flush(D, 0, D.gen - keep_plans - 1);
}
static long long be_dev_pool_trim(long long keep_bytes) {
  NmxDevPool& D = be_dev_pool();
  std::lock_guard<std::mutex> lk(D.m);
  const size_t before = D.cached;
  be_dev_pool_flush(D, keep_bytes > 0 ? (size_t)keep_bytes : 0);
  return (long long)(before - D.cached);
}
static void* be_host_alloc(size_t n) {
  void* p = nullptr;
  if (hipHostMalloc(&p, n, hipHostMallocPortable)   /* every device of a multi-device stream copies from it */ != hipSuccess) return nullptr;
  return p;
}
static void be_host_free(void* p) { (void)hipHostFree(p); }
static void be_h2d_sync(void* d, const void* s, size_t n) { BE_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); }
static void be_d2h_sync(void* d, const void* s, size_t n) { BE_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); }
static void be_memset_sync(void* d, int v, size_t n) { BE_TRY(hipMemset(d, v, n)); }
static void be_h2d_async(void* d, const void* s, size_t n, be_stream_t st) {
  BE_TRY(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st));
}
static void be_d2h_async(void* d, const void* s, size_t n, be_stream_t st) {
  BE_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st));
}
static void be_h2d_2d_async(void* d, size_t dpitch, const void* s, size_t spitch, size_t width,
                            size_t height, be_stream_t st) {
  BE_TRY(hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyHostToDevice, st));
}
static void be_d2h_2d_async(void* d, size_t dpitch, const void* s, size_t spitch, size_t width,
                            size_t height, be_stream_t st) {
  BE_TRY(hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyDeviceToHost, st));
}
static void be_memset_async(void* d, int v, size_t n, be_stream_t st) { BE_TRY(hipMemsetAsync(d, v, n, st)); }
// a host function in stream order (runs once everything enqueued on `st` before it has completed)
static void be_host_fn(be_stream_t st, void (*fn)(void*), void* arg) { BE_TRY(hipLaunchHostFunc(st, fn, arg)); }
static int be_sync(be_stream_t st) { return be_hip(hipStreamSynchronize(st), "hipStreamSynchronize"); }
// The wait at the end of a host-memory batch, with a watchdog: hipStreamSynchronize has no timeout, and a kernel that never
// ends (round 4 saw ONE such run of the knob sweep in tests/test_gpu_parity.py, never reproduced: profiles/r05_hang_soak.txt)
// leaves a caller that says nothing.  Polls the stream instead -- spinning for the first millisecond (the one-window call of a
// real-time loop returns sooner than through the blocking wait), then sleeping 50 us at a time -- and gives up after
// NMX_SYNC_TIMEOUT_S (default 300; 0: wait for ever) with the stage-by-stage kernel lists of the batch in the message.
static int be_sync_watch(be_stream_t st, const std::string* kernels, int n_lists) {
  static const double limit = [] {
    const char* v = getenv("NMX_SYNC_TIMEOUT_S");
    return (v && ((v[0] >= '0' && v[0] <= '9') || v[0] == '.')) ? atof(v) : 300.0;
  }();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return be_hip(q, "hipStreamQuery");
    (void)hipGetLastError();   // (hipErrorNotReady is sticky in the thread's last-error slot)
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (limit > 0.0 && el > limit) {
      std::string msg = "the device did not finish a batch within " + std::to_string((int)limit) + " s (NMX_SYNC_TIMEOUT_S); kernels of the last launch sequence:";
      for (int i = 1; i < n_lists; ++i)
        if (kernels && !kernels[i].empty()) msg += " [stage " + std::to_string(i) + ": " + kernels[i] + "]";
      return nmx_fail(NMX_E_HIP, msg);
    }
    if (el > 1e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (a sleep of 50 us is 100+ with timer slack: none inside a one-window call)
  }
}
static void be_sync_quiet(be_stream_t st) { if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError(); }   // (error paths: the first message stays)
// Streams of destroyed plans are recycled (per device, by priority class): hipStreamCreate is a hardware-queue creation of
// 3.7 ms and hipStreamDestroy 2.5 ms on this runtime (tools/hip_call_costs.py) -- a plan owns seven, so a fresh Stream per
// run (the reference's pattern) paid tens of milliseconds for what the device-memory pool already saves on hipMalloc.
// Only IDLE streams are kept (hipStreamQuery == success: whoever destroys a stream has synchronised it, or it never ran
// anything); at most NMX_STREAM_POOL (32, 0: off) per class and device.
struct NmxStreamPool {
  std::mutex mu;
  std::vector<hipStream_t> idle[2];                       // [0] default priority, [1] highest
  std::unordered_map<hipStream_t, int> cls;                // streams handed out by this pool -> class
};
static NmxStreamPool& be_stream_pool(int dev) {
  static std::mutex mu;
  static std::map<int, NmxStreamPool*> pools;
  std::lock_guard<std::mutex> lk(mu);
  NmxStreamPool*& p = pools[dev];
  if (!p) p = new NmxStreamPool();
  return *p;
}
static int be_stream_pool_cap() {
  static int cap = -1;
  if (cap < 0) { const char* v = getenv("NMX_STREAM_POOL"); cap = v ? std::max(0, atoi(v)) : 32; }
  return cap;
}
static be_stream_t be_stream_make(int klass) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  NmxStreamPool& P = be_stream_pool(dev);
  hipStream_t s = nullptr;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    if (!P.idle[klass].empty()) { s = P.idle[klass].back(); P.idle[klass].pop_back(); P.cls[s] = klass; return s; }
  }
  if (klass == 1) {
    int lo = 0, hi = 0;
    if (!(hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
          hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) == hipSuccess)) {
      s = nullptr;
      (void)hipGetLastError();
    }
  }
  if (!s) BE_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (s) { std::lock_guard<std::mutex> lk(P.mu); P.cls[s] = klass; }
  return s;
}
static be_stream_t be_stream_create() { return be_stream_make(0); }
static void be_stream_destroy(be_stream_t s) {
  if (!s) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  NmxStreamPool& P = be_stream_pool(dev);
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.cls.find(s);
    if (it != P.cls.end()) {
      const int klass = it->second;
      P.cls.erase(it);
      if ((int)P.idle[klass].size() < be_stream_pool_cap() && hipStreamQuery(s) == hipSuccess) { P.idle[klass].push_back(s); return; }
      (void)hipGetLastError();   // (a busy or failed stream is not kept)
    }
  }
  (void)hipStreamDestroy(s);
}
// side stream for the latency-bound bursts chain: highest priority so its few waves are issued
// ahead of the throughput kernels it overlaps with
static be_stream_t be_stream_create_high() { return be_stream_make(1); }
typedef hipEvent_t be_event_t;
static void be_event_create(be_event_t& e) { BE_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
static void be_event_destroy(be_event_t& e) { if (e) (void)hipEventDestroy(e); }
static void be_event_record(be_event_t& e, be_stream_t s) { BE_TRY(hipEventRecord(e, s)); }
static void be_stream_wait(be_stream_t s, be_event_t& e) { BE_TRY(hipStreamWaitEvent(s, e, 0)); }
static void be_timer_create(be_timer_t& t) {
  BE_TRY(hipEventCreate(&t.a));
  BE_TRY(hipEventCreate(&t.b));
}
static void be_timer_destroy(be_timer_t& t) {
  if (t.a) (void)hipEventDestroy(t.a);
  if (t.b) (void)hipEventDestroy(t.b);
}
static void be_timer_start(be_timer_t& t, be_stream_t s) { BE_TRY(hipEventRecord(t.a, s)); t.used = false; }
static void be_timer_stop(be_timer_t& t, be_stream_t s) { BE_TRY(hipEventRecord(t.b, s)); t.used = true; }
static float be_timer_elapsed(be_timer_t& t) {
  if (!t.used) return 0.f;
  float ms = 0.f;
  if (hipEventSynchronize(t.b) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) return -1.f;
  return ms;
}
static int be_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return be_hip(e, "kernel launch");
  int rc = g_be_rc;
  g_be_rc = 0;
  return rc;
}

template <typename K>
static void be_allow_lds(K kern) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
static void be_init_once() {
  static unsigned long long seen = 0;   // per device (the opt-in is a per-device attribute)
  if (!nmx_first_on_device(seen)) return;
  be_allow_lds(nmx_kern_timeosc);
  be_allow_lds(nmx_kern_bank);
  be_allow_lds(nmx_kern_hilbert);
  be_allow_lds(nmx_kern_resample);
  be_allow_lds(nmx_kern_burst_thr<32>);
  be_allow_lds(nmx_kern_burst_thr<64>);
  be_allow_lds(nmx_kern_burst_thr<128>);
  be_allow_lds(nmx_kern_burst_thr_wide);
  be_allow_lds(nmx_kern_burst_fill);
  be_allow_lds(nmx_kern_burst_fill_sort);
  be_allow_lds(nmx_kern_rawnorm_order);   // 24 * (window + hop) + 8 KiB: above 64 KiB from ~2390 samples
}

extern "C" void nmx_wave_launch_scan(const NmxTimeOscArgs* A, int n_items, hipStream_t s);
extern "C" int nmx_wave_launch_timeosc_w1000(const NmxTimeOscArgs* A, int n_items, hipStream_t s);
extern "C" int nmx_wave_launch_timeosc_w510(const NmxTimeOscArgs* A, int n_items, hipStream_t s);
extern "C" int nmx_wave_launch_timeosc_stft500(const NmxTimeOscArgs* A, int n_items, hipStream_t s);
extern "C" int nmx_specmm_launch(const NmxTimeOscArgs* A, int n_items, hipStream_t s);
extern "C" void nmx_wave_launch_timeosc_w1000_todo(const NmxTimeOscArgs* A, int n_items, hipStream_t s);
// -> 1 when the kernel launched leaves flagged windows to be_launch_timeosc_redo (the matrix-pipe kernel's dirty tiles)
static int be_launch_timeosc(const NmxTimeOscArgs& A, int n_items, int nt, size_t lds, be_stream_t s) {
  be_init_once();
  static int scan_ok = -1;
  if (scan_ok < 0) { const char* v = getenv("NMX_SCAN_KERNEL"); scan_ok = !(v && v[0] == '0'); }
  // no oscillatory feature: the register-resident scan (one wave per window, no LDS)
  if (scan_ok && !A.fft.enabled && !A.welch.enabled && !A.stft.enabled && A.W <= 1024 && A.W >= 3) {
    nmx_wave_launch_scan(&A, n_items, s);
    return 0;
  }
  // FFT band means of 1000-sample windows whose bins fit 32 rows: the spectrum on the matrix pipe (nmx_k_specmm.h)
  // (any batch size, one window included: a result must not depend on how the hops were batched)
  if (A.smm_tab) {
    const int r = nmx_specmm_launch(&A, n_items, s);
    if (r) return r == 2;
  }
  // default shape (W = 1000, band means): one wave per item, wave-level 500-point transforms
  if (A.w500_tab && nmx_wave_launch_timeosc_w1000(&A, n_items, s)) return 0;
  if (A.w500_tab && nmx_wave_launch_timeosc_stft500(&A, n_items, s)) return 0;
  // 510-sample FFT / STFT segments (17 ms at 30 kHz): one wave per item, in-place prime-factor transforms
  if (A.w510_tab && nmx_wave_launch_timeosc_w510(&A, n_items, s)) return 0;
  if (nt == 128) { nmx_timeosc_fixed_launch128(&A, n_items, lds, s); return 0; }
  hipLaunchKernelGGL(nmx_kern_timeosc, dim3(n_items), dim3(nt), lds, s, A);
  nmxi_note_kernel("nmx_kern_timeosc");
  return 0;
}
static void be_launch_timeosc_redo(const NmxTimeOscArgs& A, int n_items, be_stream_t s) {
  nmx_wave_launch_timeosc_w1000_todo(&A, n_items, s);
}
static void be_launch_bank(const NmxBankArgs& A, int n_items, int nt, size_t lds, be_stream_t s) {
  be_init_once();
  const int grid = (A.partitioned && n_items > NMX_UPS_SLOTS) ? NMX_UPS_SLOTS : n_items;
  hipLaunchKernelGGL(nmx_kern_bank, dim3(grid), dim3(nt), lds, s, A, n_items);
  nmxi_note_kernel("nmx_kern_bank");
}
extern "C" void nmx_w64_launch_rd64(const NmxBankW64Args*, int, size_t, hipStream_t);
extern "C" int nmx_w64p_launch_rd64(const NmxBankW64Args*, int, int, hipStream_t);
extern "C" int nmx_w64q_launch_notch_rd64(const NmxBankW64Args*, int, hipStream_t);
extern "C" int nmx_w64x2_launch_rd64(const NmxBankW64Args*, int, int, hipStream_t);
extern "C" int nmx_w64c_launch_rd64(const NmxBankW64Args*, int, int, hipStream_t);
extern "C" int nmx_w64d_launch_rd64(const NmxBankW64Args*, int, int, hipStream_t);
extern "C" int nmx_w64e_launch_rd64(const NmxBankW64Args*, int, int, hipStream_t);
extern "C" void nmx_wave_launch_sharp_todo(const NmxSharpArgs* A, int n_items, size_t lds, const unsigned char* todo,
                                           hipStream_t s);
// One-wave FIR kernels (nmx_w64.hip), one kernel per shape class:
//   M = 4096 (windows + filter half-length in (2048, 4096])            nmx_kern_bank_w64x2
//   channel pairs, M = 1536 / 1024 (every filter that fits)            nmx_kern_bank_w64c / w64d
//   channel pairs, M = 2048 (longer filters; the notch)                nmx_kern_bank_w64e<0 / 1>
//   M = 2048, >= 4096 items: persistent 8-wave workgroups, pipelined   nmx_kern_bank_w64pp
//   notch (odd-reflected window), >= 1024 items: four items / workgroup nmx_kern_notch_w64q
//   a window or two (nmx_process_window): one wave per workgroup        nmx_kern_bank_w64 / nmx_kern_notch_w64
static void be_launch_bank_w64(const NmxBankW64Args& A, int n_items, size_t lds, be_stream_t s) {
  static thread_local int n_cu = -1, n_cu_dev = -1;   // per host thread and device (multi-device streams)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (n_cu < 0 || n_cu_dev != dev) {
    hipDeviceProp_t prop;
    n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    n_cu_dev = dev;
  }
  if (A.tw2) {
    if (!nmx_w64x2_launch_rd64(&A, n_items, n_cu, s)) g_be_rc = nmx_fail(NMX_E_INVALID, "M = 4096 FIR path: LDS budget");
    return;
  }
  if (A.hc && A.pair_m == 2048) {
    if (nmx_w64e_launch_rd64(&A, n_items, n_cu, s)) return;   // else: the one-channel M = 2048 kernels below
  } else if (A.hc && (A.pair_m == 1024 ? nmx_w64d_launch_rd64(&A, n_items, n_cu, s) : nmx_w64c_launch_rd64(&A, n_items, n_cu, s))) return;
  if (n_items >= 4096 && nmx_w64p_launch_rd64(&A, n_items, n_cu, s)) return;
  if (A.b.pad_mode != 0 && n_items >= 1024 && nmx_w64q_launch_notch_rd64(&A, n_items, s)) return;
  nmx_w64_launch_rd64(&A, n_items, lds, s);
}
extern "C" int nmx_w64_takes_dc_rd64(const NmxBankW64Args*, int);
static bool be_bank_w64_takes_dc(const NmxBankW64Args& A, int n_items) { return nmx_w64_takes_dc_rd64(&A, n_items) != 0; }
extern "C" int nmx_wave_timeosc_takes_dc(const NmxTimeOscArgs* A);
static bool be_timeosc_takes_dc(const NmxTimeOscArgs& A) { return nmx_wave_timeosc_takes_dc(&A) != 0; }
extern "C" void nmx_wave_launch_sharp_dense(const NmxSharpArgs* A, int n_items, hipStream_t s);
static void be_launch_sharp_dense(const NmxSharpArgs& A, int n_items, be_stream_t s) {
  be_init_once();
  nmx_wave_launch_sharp_dense(&A, n_items, s);
}
static void be_launch_sharp_todo(const NmxSharpArgs& A, int n_items, size_t lds, const unsigned char* todo, be_stream_t s) {
  be_init_once();
  nmx_wave_launch_sharp_todo(&A, n_items, lds, todo, s);
}
extern "C" void nmx_wave_launch_hilbert_w500(const NmxHilbertArgs* A, long long n_items, hipStream_t s);
extern "C" void nmx_wave_launch_hilbert_w1000(const NmxHilbertArgs* A, long long n_items, hipStream_t s);
static void be_launch_hilbert(const NmxHilbertArgs& A, long long n_items, int nt, size_t lds, be_stream_t s) {
  be_init_once();
  static int w500 = -1;
  if (w500 < 0) { const char* v = getenv("NMX_HILBERT_W500"); w500 = !(v && v[0] == '0'); }
  if (w500 && A.W == 1000 && !A.hil_full) { nmx_wave_launch_hilbert_w500(&A, n_items, s); return; }
  if (w500 && A.W == 2000 && A.w1000_tab) { nmx_wave_launch_hilbert_w1000(&A, n_items, s); return; }
  if (nt == 128) { nmx_hilbert_fixed_launch128(&A, n_items, lds, s); return; }
  hipLaunchKernelGGL(nmx_kern_hilbert, dim3((unsigned)n_items), dim3(nt), lds, s, A);
  nmxi_note_kernel("nmx_kern_hilbert");
}
extern "C" void nmx_wave_launch_burst_thr(const NmxBurstThrArgs* A, int n_items, hipStream_t s, long long windows_seen);
// windows_seen: hops every sequence has absorbed before this batch (-1: always the workgroup kernel)
static void be_launch_burst_thr(const NmxBurstThrArgs& A, int n_items, int nt, size_t lds, be_stream_t s,
                                long long windows_seen = -1) {
  be_init_once();
  // ring already full at the first hop: the barrier-free one-wave walk over the list in L2
  if (windows_seen > 0 && nmx_burst_thr_wave_ok(A, windows_seen)) { nmx_wave_launch_burst_thr(&A, n_items, s, windows_seen); return; }
  const int chunk = (A.K + nt - 1) / nt;
  if (nt > 256) { hipLaunchKernelGGL(nmx_kern_burst_thr_wide, dim3(n_items), dim3(1024), lds, s, A); nmxi_note_kernel("nmx_kern_burst_thr_wide"); }
  else if (chunk <= 32) { hipLaunchKernelGGL(nmx_kern_burst_thr<32>, dim3(n_items), dim3(nt), lds, s, A); nmxi_note_kernel("nmx_kern_burst_thr<32>"); }
  else if (chunk <= 64) { hipLaunchKernelGGL(nmx_kern_burst_thr<64>, dim3(n_items), dim3(nt), lds, s, A); nmxi_note_kernel("nmx_kern_burst_thr<64>"); }
  else { hipLaunchKernelGGL(nmx_kern_burst_thr<128>, dim3(n_items), dim3(nt), lds, s, A); nmxi_note_kernel("nmx_kern_burst_thr<128>"); }
}
static void be_launch_burst_fill(const NmxBurstThrArgs& A, int n_items, unsigned short* slots, float* sorted, be_stream_t s) {
  be_init_once();
  int n2 = 2048;
  while (n2 < A.W + (A.n_windows - 1) * A.overlap) n2 <<= 1;
  const char* v_split = getenv("NMX_FILL_SPLIT");   // (read per launch -- once per fresh stream: the tests run both forms in one process)
  const bool split = !(v_split && v_split[0] == '0');
  const size_t lds_walk = nmx_burst_fill_walk_lds(n2, A.n_windows);
  if (split && lds_walk <= 48 * 1024) {   // (a hop count whose slot pairs do not fit: the one-launch form)
    hipLaunchKernelGGL(nmx_kern_burst_fill_sort, dim3(n_items), dim3(NMX_FILL_NT), nmx_burst_fill_sort_lds(n2), s, A, n2, slots, sorted);
    hipLaunchKernelGGL(nmx_kern_burst_fill_walk, dim3(n_items), dim3(64), lds_walk, s, A, n2, (const unsigned short*)slots,
                       (const float*)sorted);
    nmxi_note_kernel("nmx_kern_burst_fill_sort");
    nmxi_note_kernel("nmx_kern_burst_fill_walk");
    return;
  }
  hipLaunchKernelGGL(nmx_kern_burst_fill, dim3(n_items), dim3(NMX_FILL_NT), nmx_burst_fill_lds(n2), s, A, n2, slots);
  nmxi_note_kernel("nmx_kern_burst_fill");
}
static void be_launch_burst_stat(const NmxBurstStatArgs& A, int n_items, size_t lds, be_stream_t s) {
  be_init_once();
  nmx_wave_launch_burst_stat(&A, n_items, lds, s);
}
static void be_launch_sharp(const NmxSharpArgs& A, int n_items, size_t lds, be_stream_t s) {
  be_init_once();
  nmx_wave_launch_sharp(&A, n_items, lds, s);
}
static void be_launch_reref(const NmxRerefArgs& A, be_stream_t s) {
  dim3 grid((unsigned)((A.T + 255) / 256), (unsigned)((A.C + NMX_REREF_ROWS - 1) / NMX_REREF_ROWS));
  hipLaunchKernelGGL(nmx_kern_reref, grid, dim3(256), 0, s, A);
  nmxi_note_kernel("nmx_kern_reref");
}
static void be_launch_resample(const NmxResampleArgs& A, int n_items, int nt, size_t lds, be_stream_t s) {
  be_init_once();
  hipLaunchKernelGGL(nmx_kern_resample, dim3(n_items), dim3(nt), lds, s, A);
  nmxi_note_kernel("nmx_kern_resample");
}
static void be_launch_rawnorm(const NmxRawNormArgs& A, be_stream_t s) {
  be_init_once();
  if (A.method >= NMX_RAWNORM_MEDIAN && A.method != NMX_RAWNORM_POWER) {
    // lists in LDS + reduction scratch, or (lists in device memory) reduction scratch + subsample scratch
    const size_t lds = A.lists ? (size_t)8 * NMX_RAWNORM_ORDER_NT + 4 * (2 * NMX_RAWNORM_ORDER_NT + 272)
                               : (size_t)24 * A.max_list + 8 * NMX_RAWNORM_ORDER_NT;
    hipLaunchKernelGGL(nmx_kern_rawnorm_order, dim3(A.n_channels), dim3(NMX_RAWNORM_ORDER_NT), lds, s, A);
    const long long n = (long long)A.n_windows * A.n_channels * A.W;
    hipLaunchKernelGGL(nmx_kern_rawnorm_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A);
    nmxi_note_kernel("nmx_kern_rawnorm_order + nmx_kern_rawnorm_apply");
    return;
  }
  hipLaunchKernelGGL(nmx_kern_rawnorm_stats, dim3(A.n_channels), dim3(64), 0, s, A);
  const long long n = (long long)A.n_windows * A.n_channels * A.W;
  hipLaunchKernelGGL(nmx_kern_rawnorm_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A);
  nmxi_note_kernel("nmx_kern_rawnorm_stats + nmx_kern_rawnorm_apply");
}
static void be_launch_kalman(const NmxKalmanArgs& A, be_stream_t s) {
  const int n = A.n_channels * A.n_bands;
  hipLaunchKernelGGL(nmx_kern_kalman, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, A);
  nmxi_note_kernel("nmx_kern_kalman");
}
static void be_launch_norm(const NmxNormArgs& A, be_stream_t s) {
  // one thread per column, one wave per workgroup: columns spread over as many CUs as possible
  hipLaunchKernelGGL(nmx_kern_norm, dim3((unsigned)((A.n_cols + 63) / 64)), dim3(64), 0, s, A);
}
static void be_launch_norm_scan(const NmxNormArgs& A, const NmxNormScan& S, be_stream_t s) {
  const unsigned cb = (unsigned)((A.n_cols + 63) / 64);
  const dim3 cells((unsigned)A.n_rows, (unsigned)((A.n_cols + 255) / 256));
  if (S.n_hseg) hipLaunchKernelGGL(nmx_kern_norm_seg_hist, dim3(cb, (unsigned)S.n_hseg), dim3(64), 0, s, A, S);
  hipLaunchKernelGGL(nmx_kern_norm_seg_batch, dim3(cb, (unsigned)S.n_bseg), dim3(64), 0, s, A, S);
  hipLaunchKernelGGL(nmx_kern_norm_seg_offsets, dim3(cb), dim3(64), 0, s, A, S);
  hipLaunchKernelGGL(nmx_kern_norm_cell, cells, dim3(256), 0, s, A, S);
  hipLaunchKernelGGL(nmx_kern_norm_ring, cells, dim3(256), 0, s, A, S);
}
static void be_launch_power(const NmxPowerPrepArgs& P, const NmxPowerArgs& A, be_stream_t s) {
  const unsigned gy = (unsigned)((P.n_cols + 255) / 256);
  hipLaunchKernelGGL(nmx_kern_power_prep, dim3((unsigned)(P.have + P.n_rows), gy), dim3(256), 0, s, P);
  // one wave per 64 columns of one hop: the ~30 likelihood evaluations of a fit diverge little inside a wave
  hipLaunchKernelGGL(nmx_kern_power, dim3((unsigned)A.n_rows, (unsigned)((A.n_cols + 63) / 64)), dim3(64), 0, s, A);
  hipLaunchKernelGGL(nmx_kern_power_ring, dim3((unsigned)P.n_rows, gy), dim3(256), 0, s, P);
}
static void be_launch_car(const NmxCarArgs& A, be_stream_t s) {
  if (A.T <= 4096 && A.C >= 64) hipLaunchKernelGGL(nmx_kern_car<16>, dim3((unsigned)((A.T + 63) / 64)), dim3(1024), 0, s, A);
  else hipLaunchKernelGGL(nmx_kern_car<4>, dim3((unsigned)((A.T + 63) / 64)), dim3(256), 0, s, A);
  nmxi_note_kernel("nmx_kern_car");
}
static void be_launch_reref_struct(const NmxRerefStructArgs& A, be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_reref_struct, dim3((unsigned)((A.T + 63) / 64)), dim3(256), 0, s, A);
  nmxi_note_kernel("nmx_kern_reref_struct");
}
static void be_launch_shift(const NmxShiftArgs& A, be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_shift, dim3((unsigned)((A.T + 255) / 256), (unsigned)A.C), dim3(256), 0, s, A);
  nmxi_note_kernel("nmx_kern_shift");
}
__global__ void __launch_bounds__(256) nmx_kern_reref64(const NmxReref64Args A) {
  nmx_reref64_tile(A, (long long)blockIdx.x * 256 + threadIdx.x, (int)blockIdx.y * NMX_REREF64_ROWS);
}
static void be_launch_reref64(const NmxReref64Args& A, be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_reref64, dim3((unsigned)((A.T + 255) / 256), (unsigned)((A.C + NMX_REREF64_ROWS - 1) / NMX_REREF64_ROWS)),
                     dim3(256), 0, s, A);
}
__global__ void __launch_bounds__(256) nmx_kern_rs64_elem(const NmxResample64Args A, int mode, const NmxCplx64* src, NmxCplx64* dst) {
  nmx_rs64_elem(A, mode, src, dst, (int)blockIdx.y, (long long)blockIdx.x * 256 + threadIdx.x);
}
__global__ void __launch_bounds__(256) nmx_kern_rs64_pass(const NmxCplx64* src, NmxCplx64* dst, long long ld, long long n, long long ns,
                                                          long long st, int sign) {
  nmx_rs64_pass(src, dst, ld, n, ns, st, sign, (int)blockIdx.y, (long long)blockIdx.x * 256 + threadIdx.x);
}
static void be_launch_rs64_elem(const NmxResample64Args& A, int mode, const NmxCplx64* src, NmxCplx64* dst, long long n, int rows,
                                be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_rs64_elem, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, s, A, mode, src, dst);
}
static void be_launch_rs64_pass(const NmxCplx64* src, NmxCplx64* dst, long long ld, long long n, long long ns, long long st, int sign,
                                int rows, be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_rs64_pass, dim3((unsigned)((n / 2 + 255) / 256), (unsigned)rows), dim3(256), 0, s, src, dst, ld, n, ns,
                     st, sign);
}
static void be_launch_nanmask(const NmxNanMaskArgs& A, int n_items, be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_nanmask, dim3(n_items), dim3(64), 64 * sizeof(float), s, A);
}
static void be_launch_tap(const NmxTapArgs& A, int n_items, be_stream_t s) {
  hipLaunchKernelGGL(nmx_kern_tap, dim3(n_items), dim3(256), 0, s, A);
}

#include "nmx_engine.inc"

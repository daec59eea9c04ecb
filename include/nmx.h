/*
 * nmx.h -- C ABI of libnmx.so, the MI355X (gfx950) engine for py_neuromodulation's per-hop
 * hot path  nm.Stream -> DataProcessor.process -> filter/ -> features/.
 *
 * The reference is pure Python and has NO FFI for this path; the plugin seam it offers is
 *   NMFeature.__init__(settings, ch_names, sfreq) / calc_feature(data[C, W]) -> dict
 *       (py_neuromodulation/utils/types.py:59-77)
 *   NMPreprocessor.process(data[C, W]) -> data        (utils/types.py:80-81)
 *   DataProcessor.process(data[C_all, W]) -> dict     (stream/data_processor.py:238-311)
 * so the entry points below are what a ctypes binding inside those three call shapes needs
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative NMX_E_* code; nmx_last_error()
 *     returns a thread-local message for the last failure on the calling thread.
 *   - the caller owns every buffer it passes in; the plan owns device scratch, FIR tap
 *     spectra, FFT twiddles and the burst state; nmx_plan_destroy frees them.
 *   - one plan = one logical stream on one GPU, not re-entrant (the reference calls
 *     process() from a single thread, stream/stream.py:280-296); distinct plans are
 *     independent (one per GPU for channel sharding).
 *   - there is NO CPU fallback: without a HIP device nmx_plan_create fails with
 *     NMX_E_NODEVICE.
 */
#ifndef NMX_H
#define NMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libnmx.so is built with -fvisibility=hidden: the prototypes below are its whole link-visible surface */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define NMX_ABI_VERSION 11

/* error codes */
#define NMX_OK 0
#define NMX_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define NMX_E_NODEVICE (-2)  /* no HIP device (the engine has no CPU path) */
#define NMX_E_HIP (-3)       /* a HIP runtime call failed */
#define NMX_E_NOMEM (-4)

/* feature bits of nmx_plan_desc.features; order = FeatureSelector field order
 * (stream/settings.py:41-55), which is the reference's execution and column order */
#define NMX_F_HJORTH (1u << 0)     /* features/hjorth_raw.py:18-42   */
#define NMX_F_RAW (1u << 1)        /* features/hjorth_raw.py:45-57   */
#define NMX_F_BANDPOWER (1u << 2)  /* features/bandpower.py:98-207   */
#define NMX_F_STFT (1u << 3)       /* features/oscillatory.py:185-250 */
#define NMX_F_FFT (1u << 4)        /* features/oscillatory.py:58-119  */
#define NMX_F_WELCH (1u << 5)      /* features/oscillatory.py:122-182 */
#define NMX_F_SHARPWAVE (1u << 6)  /* features/sharpwaves.py:100-465  */
#define NMX_F_BURSTS (1u << 7)     /* features/bursts.py:60-298       */
#define NMX_F_LINELENGTH (1u << 8) /* features/linelength.py:11-21    */

/* estimator bits (oscillatory: mean/median/std/max, features/oscillatory.py:29-34) */
#define NMX_EST_MEAN 1u
#define NMX_EST_MEDIAN 2u
#define NMX_EST_STD 4u
#define NMX_EST_MAX 8u

/* sharp-wave per-trough quantities, SharpwaveFeatures field order (sharpwaves.py:43-56) */
enum {
  NMX_SW_PEAK_LEFT = 0, NMX_SW_PEAK_RIGHT, NMX_SW_NUM_PEAKS, NMX_SW_TROUGH, NMX_SW_WIDTH,
  NMX_SW_PROMINENCE, NMX_SW_INTERVAL, NMX_SW_DECAY_TIME, NMX_SW_RISE_TIME, NMX_SW_SHARPNESS,
  NMX_SW_RISE_STEEPNESS, NMX_SW_DECAY_STEEPNESS, NMX_SW_SLOPE_RATIO, NMX_SW_NFEAT
};
/* sharp-wave estimators, ESTIMATOR_DICT order (sharpwaves.py:26-32) */
enum { NMX_SWE_MEAN = 0, NMX_SWE_MEDIAN, NMX_SWE_MAX, NMX_SWE_MIN, NMX_SWE_VAR, NMX_SWE_N };

#define NMX_MAX_BANDS 16
#define NMX_MAX_FILTERS 24
#define NMX_MAX_SW_COMBOS 48
#define NMX_MAX_PRE_FILTERS 4

/* Where one feature family writes inside an output row of n_outputs floats:
 *   column = base + ch * ch_stride + a * a_stride + b * b_stride
 * (a, b) are family specific, see each family below.  This reproduces the reference's key
 * order without any host-side permutation. */
typedef struct {
  int32_t base, ch_stride, a_stride, b_stride;
} nmx_cols;

/* oscillatory family (FFT / Welch / STFT): band b uses bins [bin_lo[b], bin_hi[b]) of the
 * family's frequency grid (the host resolves [lo,hi) vs [lo,hi] -- oscillatory.py:81,206).
 * cols: a = band, b = estimator slot (enabled estimators in mean,median,std,max order).
 * psd_cols (return_spectrum): a = frequency bin, used when return_spectrum != 0. */
typedef struct {
  int32_t n;               /* FFT: samples N = floor(ms/1000*sfreq); Welch: nperseg; STFT: nperseg */
  int32_t log_transform;
  uint32_t estimators;     /* NMX_EST_* */
  int32_t return_spectrum;
  int32_t bin_lo[NMX_MAX_BANDS], bin_hi[NMX_MAX_BANDS];
  nmx_cols cols, psd_cols;
} nmx_osc_desc;

/* One FIR filter of the per-window bank (band-pass bank of BandPower/Bursts,
 * filter/mne_filter.py:35-128, and the sharp-wave pre-filters, sharpwaves.py:121-154).
 * Taps are DESIGNED ON THE HOST and passed in (odd length, float64). */
typedef struct {
  const double* taps;
  int32_t n_taps;
  /* BandPower epilogue (bandpower.py:130-207): tail length in samples, 0 = not used.
   * cols: a = feature slot among enabled (activity, mobility, complexity). */
  int32_t bp_seglen;
  int32_t bp_band_index;    /* band position for the output column */
  /* Bursts epilogue: index among burst bands or -1 */
  int32_t burst_index;
  /* Sharp-wave epilogue: index among sharp-wave filters or -1 */
  int32_t sw_index;
} nmx_filter_desc;

typedef struct {
  int32_t abi_version;      /* NMX_ABI_VERSION */
  int32_t device;           /* HIP device ordinal */
  int32_t n_channels;       /* C: channels the features are computed for */
  int32_t window;           /* W: samples per window (int(seg_ms/1000*sfreq)) */
  double sfreq;
  double feat_hz;           /* sampling_rate_features_hz (burst ring: samples_overlap) */
  uint32_t features;        /* NMX_F_* */
  int32_t n_outputs;        /* floats per output row (all enabled families) */
  int32_t n_bands;

  nmx_cols hjorth_cols;     /* a = 0..2 (Activity, Mobility, Complexity) */
  nmx_cols raw_cols;
  nmx_cols linelength_cols;
  nmx_osc_desc fft, welch, stft;

  /* FIR bank */
  int32_t n_filters;
  nmx_filter_desc filters[NMX_MAX_FILTERS];
  uint32_t bp_features;     /* bit0 activity, bit1 mobility, bit2 complexity */
  int32_t bp_log_transform;
  nmx_cols bp_cols;         /* a = band, b = feature slot */

  /* Bursts (features/bursts.py): cols a = burst band, b = output slot in the order
   * duration_mean, duration_max, amplitude_mean, amplitude_max, burst_rate_per_s, in_burst
   * restricted to the enabled groups (burst_out_mask bit i = slot i present). */
  int32_t n_burst_bands;
  double burst_threshold;   /* percentile 0..100 */
  double burst_time_duration_s;
  uint32_t burst_out_mask;
  nmx_cols burst_cols;

  /* Sharp waves (features/sharpwaves.py) */
  int32_t n_sw_filters;
  int32_t sw_n_combos;                     /* (feature, estimator) pairs, reference order */
  int32_t sw_combo_feature[NMX_MAX_SW_COMBOS];
  int32_t sw_combo_estimator[NMX_MAX_SW_COMBOS];
  double sw_distance_peaks, sw_distance_troughs;  /* samples (ms passed as samples, :339-344) */
  int32_t sw_estimate_peaks, sw_estimate_troughs; /* detect_peaks/troughs.estimate */
  int32_t sw_between;                      /* apply_estimator_between_peaks_and_troughs */
  nmx_cols sw_cols;        /* a = filter, b = combo (x2 + polarity when !sw_between) */
  nmx_cols sw_numpeaks_cols; /* a = filter; used when num_peaks is enabled and sw_between */

  /* Pre-processing on the continuous stream / per window */
  const double* notch_taps; /* NULL = no notch; filter/notch_filter.py:9-93 */
  int32_t n_notch_taps;
  const double* ref_matrix; /* NULL = identity; [C][C_in] row-major, processing/rereference.py:52-100 */
  int32_t n_channels_in;    /* rows of the incoming data when ref_matrix is given */

  /* Kalman smoothing of bandpass_activity (features/bandpower.py:147-163,188-189;
   * filter/kalman_filter.py:45-78): white-noise-acceleration model, one 2-state filter per
   * (channel, band) with bit `band` of bp_kalman_mask set, predict + update once per hop on the
   * (log-)activity BEFORE nan_to_num.  0 = off.  State is part of nmx_state_*. */
  uint32_t bp_kalman_mask;
  double kalman_Tp, kalman_sigma_w, kalman_sigma_v;

  /* raw_resampling (processing/resample.py:19-60 -> mne.filter.resample, FFT method): every
   * incoming window of raw_window samples is resampled by resample_ratio = new_sfreq / old_sfreq to
   * `window` = round(ratio * raw_window) samples before notch / features; `sfreq` is the NEW rate.
   * raw_window == 0: no resampling (incoming windows have `window` samples). */
  int32_t raw_window;
  double resample_ratio;

  /* preprocessing_filter (processing/filter_preprocessing.py:44-94): up to NMX_MAX_PRE_FILTERS
   * single FIRs applied one after the other to every incoming window, each as
   * MNEFilter.filter_data = zero-padded "same" convolution (filter/mne_filter.py:82-128), BEFORE
   * the notch (processing/data_preprocessor.py:9-15).  Taps designed on the host, odd length. */
  int32_t n_pre_filters;
  const double* pre_taps[4];
  int32_t n_pre_taps[4];

  /* raw_normalization (processing/normalization.py:31-116, type "raw"), the last pre-processor:
   * method 0 = off, 1 = mean ((x - mean) / mean), 2 = zscore ((x - mean) / std, std 0 -> 1);
   * statistics per channel over the first window + the last raw_norm_add = int(sfreq / feat_hz)
   * samples of every later window, history trimmed to raw_norm_n - 1 samples
   * (raw_norm_n = int(normalization_time_s * sfreq)); clip <= 0: none.  Stateful (nmx_state_*). */
  int32_t raw_norm_method;
  int32_t raw_norm_n, raw_norm_add;
  float raw_norm_clip;

  /* settings.segment_length_features_ms / 1000 as the reference's Bursts uses it (features/bursts.py:81-85:
   * samples_overlap = int(sfreq * seg_s / feat_hz), burst_rate_per_s = duration_mean / seg_s).  0 = derive
   * it as window / sfreq.  It differs from window / sfreq when raw_resampling changes the window length
   * while the features are still built with the RAW rate (stream/data_processor.py:55,68,80). */
  double segment_length_s;
} nmx_plan_desc;

typedef struct nmx_plan nmx_plan;

int nmx_abi_version(void);
int nmx_device_count(void);                 /* number of HIP devices (0 when none) */
const char* nmx_last_error(void);           /* thread-local message of the last failure */

int nmx_plan_create(const nmx_plan_desc* desc, nmx_plan** out);
int nmx_plan_destroy(nmx_plan* plan);
int nmx_plan_n_outputs(const nmx_plan* plan, int64_t* n_outputs);

/* Offset split of the fp32 path.  The reference computes in float64 from the raw recording
 * (stream/data_processor.py:238-260): a channel's DC offset of 10^3 .. 10^5 times its signal costs it nothing, a cast to
 * float32 rounds at the offset's magnitude.  A plan therefore accepts a recording as  x[j][t] = u[j][t] + d[j]:
 *   nmx_plan_set_offsets   d_in[n_channels_in] float64 (copied; NULL = none): every x handed to nmx_process_batch /
 *                          _batch_tap AFTERWARDS is u = x - d, formed by the caller in float64 BEFORE its cast;
 *                          nmx_process_window / nmx_preprocess_window (float64 in) subtract d themselves.  The constants
 *                          travel through the linear stages in float64 (re-reference: R d; notch: d * sum of taps), the
 *                          features that see a constant (Raw, bin 0 of the FFT, the STFT's window share, the zero-padded
 *                          FIR bank) get it back on the device; results are those of the recording x.
 *                          NMX_E_INVALID for a plan that cannot carry offsets (nmx_plan_carries_offsets: a resampler, a
 *                          raw normaliser, a preprocessing filter or a notch longer than the window is not affine in
 *                          the window the way the split needs).
 *   without host offsets   float32 input in front of a re-reference is split by the library: the mean of a row over the first
 *                          window the plan ever sees is that row's constant when it exceeds four times the row's spread
 *                          there (learned once; part of the state of nmx_state_*,
 *                          cleared by nmx_state_reset), subtracted where the re-reference kernel loads the sample.
 *   nmx_plan_get_offsets   d_in[n_channels_in] = host + learned constants of the input rows, d_pre[n_channels] = offset of
 *                          the PRE-PROCESSED windows (what nmx_process_batch_tap's float32 windows have to be raised by),
 *                          *state = bit 0 host offsets set | bit 1 constants learned; any pointer may be NULL. */
/* Hand-shakes of a HOST-memory batch (memspace 0) with conversion threads of the caller, so that converting the recording to
 * float32 and the feature rows to whatever the caller wants runs NEXT TO the copies and kernels instead of around them:
 *   in_ready_samples   (may be NULL) the caller's counter: samples [0, *in_ready) of every row of x are in place.  The call
 *                      waits, chunk by chunk, until the samples a chunk reads are covered before it enqueues their copy.
 *   out_done_windows   (may be NULL) the library's counter: rows [0, *out_done) of `out` (and of the NaN mask) have landed
 *                      in the caller's buffers; n_windows when the call returns.
 * Both are read / written with acquire / release atomics; they stay registered until replaced (NULL, NULL = off). */
int nmx_plan_set_pipeline(nmx_plan* plan, const volatile int64_t* in_ready_samples, volatile int64_t* out_done_windows);

int nmx_plan_carries_offsets(const nmx_plan* plan, int* yes);
int nmx_plan_set_offsets(nmx_plan* plan, const double* d_in);
int nmx_plan_get_offsets(nmx_plan* plan, double* d_in, double* d_pre, int* state);

/* Batch of windows over a continuous recording x[C_in][T] (row-major, channel stride ldx),
 * window i = samples [starts[i], starts[i] + W)  (stream/generator.py:41-53).
 *   memspace 0: x / out are host pointers (copied through the plan's staging buffers)
 *   memspace 1: x / out / nan_mask are device pointers on the plan's device
 * out[n_windows][n_outputs] float32; nan_mask[n_windows][C_in] uint8 (may be NULL):
 * 1 where the raw window of that channel held a NaN (stream/data_processor.py:253).
 * hip_stream: hipStream_t to launch on (NULL = the plan's own stream).  The call is
 * asynchronous for memspace 1 and synchronous for memspace 0.
 * Bursts state (ring of envelopes) advances by n_windows. */
int nmx_process_batch(nmx_plan* plan, const float* x, int64_t ldx, int64_t n_samples,
                      const int64_t* starts, int64_t n_windows, float* out, uint8_t* nan_mask,
                      int memspace, void* hip_stream);

/* nmx_process_batch that ALSO hands back the pre-processed windows the features were computed from:
 * pre[n_windows][n_channels][window] float32 (host or device like `out`) -- the `data` argument of
 * NMFeature.calc_feature (features/feature_processor.py:80-82).  The Python host runs user-registered features
 * (nm.add_custom_feature, feature_processor.py:52-53,90-108) on them and appends their columns after the built-in
 * ones.  State (burst ring, Kalman, raw normaliser) advances exactly as in nmx_process_batch: the windows come
 * from the SAME pre-processing pass, not from a second one. */
int nmx_process_batch_tap(nmx_plan* plan, const float* x, int64_t ldx, int64_t n_samples,
                          const int64_t* starts, int64_t n_windows, float* out, uint8_t* nan_mask,
                          int memspace, void* hip_stream, float* pre);

/* One window, the reference's call shape: x[C_in][W] float64 host -> out[n_outputs] host. */
int nmx_process_window(nmx_plan* plan, const double* x, int64_t ldx, float* out,
                       uint8_t* nan_mask);

/* Pre-processing only (NMPreprocessor.process): x[C_in][W_in] float64 host -> y[C][W] float64
 * (W_in = raw_window when the plan resamples, else W). */
int nmx_preprocess_window(nmx_plan* plan, const double* x, int64_t ldx, double* y, int64_t ldy);

/* FIR bank only (MNEFilter.filter_data): x[C][W] -> y[C][n_filters][W] float64 host. */
int nmx_filter_window(nmx_plan* plan, const double* x, int64_t ldx, double* y);

/* State carried across windows: burst histories (features/bursts.py:105-115), Kalman filters, raw-normaliser sample
 * histories.  The blob is opaque and belongs to one library build; a plan built from the same description with ANOTHER
 * window length accepts it (ragged window lengths of a non-integer sampling rate: the raw-normaliser part carries the
 * exporting plan's ring capacity and is re-laid on import), so n_bytes of nmx_state_import is the exporter's
 * nmx_state_size. */
int nmx_state_reset(nmx_plan* plan);
int nmx_state_size(const nmx_plan* plan, int64_t* n_bytes);
int nmx_state_export(nmx_plan* plan, void* dst, int64_t n_bytes);
int nmx_state_import(nmx_plan* plan, const void* src, int64_t n_bytes);

/* Timing of the last nmx_process_batch, measured with HIP events on the launch stream:
 * which = 0 whole batch, 1 pre-processing, 2 time/oscillatory kernel, 3 FIR-bank kernel,
 * 4 bursts kernels, 5 sharp-wave kernel, 6 second FIR-bank launch (the filters whose taps are too long for the
 * M = 1536 channel-pair kernel; 0 when there is none).  Blocks until the events have completed. */
int nmx_last_timing_ms(nmx_plan* plan, int which, float* ms);

/* Names of the kernels the first launch sequence of the last nmx_process_batch ran in stage `which`
 * (1..6 as above; several kernels are joined by " + "), spelled as rocprofv3 --kernel-trace prints them
 * (template arguments included).  Which variant runs depends on the shape, the batch size and the tuning
 * knobs, so measurement code names the kernel from here instead of hard-coding it.  NUL-terminated,
 * truncated to n - 1 characters. */
int nmx_last_kernels(nmx_plan* plan, int which, char* buf, int64_t n);

/* ---- Feature normalisation over a batch of hops (processing/normalization.py:31-111,150-163) ----
 * The reference post-processes every feature vector with a rolling normaliser (on by default,
 * default_settings.yaml:69-78).  One nmx_norm carries the history of one stream.
 *   method      NMX_NORM_MEAN ((x - mean) / mean), NMX_NORM_ZSCORE ((x - mean) / std, std 0 -> 1),
 *               NMX_NORM_MEDIAN or NMX_NORM_ZSCORE_MEDIAN (the same with the NaN-ignoring median);
 *               statistics over the last n_hist rows INCLUDING the current one, NaNs ignored
 *   clip        > 0: clip to [-clip, clip]; <= 0: none          (normalization.py:104-105)
 *   n_hist      int(normalization_time_s * sampling_rate_features_hz) >= 2
 *   colmask     optional [n_cols] uint8, 0 = column is passed through ("psd" keys when
 *               normalize_psd is false, stream/data_processor.py:263-290); copied
 * nmx_norm_process normalises rows[n_rows][ld] IN PLACE, hop by hop semantics (row i sees rows
 * < i of the same batch and the carried history); the first row ever seen is returned unchanged.
 * memspace / hip_stream as in nmx_process_batch. */
#define NMX_NORM_MEAN 0
#define NMX_NORM_ZSCORE 1
#define NMX_NORM_MEDIAN 2          /* (x - median) / median        (normalization.py:155-157) */
#define NMX_NORM_ZSCORE_MEDIAN 3   /* (x - median) / std, std 0 -> 1 (normalization.py:166-169) */
/* the scikit-learn based methods (normalization.py:57-70,172-186): the scaler is FITTED on nan_to_num(history) every hop */
#define NMX_NORM_ROBUST 4          /* RobustScaler */
#define NMX_NORM_MINMAX 5          /* MinMaxScaler */
#define NMX_NORM_QUANTILE 6        /* QuantileTransformer(n_quantiles = 300) */
#define NMX_NORM_POWER 7           /* PowerTransformer (Yeo-Johnson, lambda by maximum likelihood, standardised) */
typedef struct nmx_norm nmx_norm;
int nmx_norm_create(int32_t device, int32_t n_cols, int32_t method, float clip, int32_t n_hist,
                    const uint8_t* colmask, nmx_norm** out);
int nmx_norm_destroy(nmx_norm* norm);
int nmx_norm_process(nmx_norm* norm, float* rows, int64_t ld, int64_t n_rows, int memspace,
                     void* hip_stream);
int nmx_norm_reset(nmx_norm* norm);
int nmx_norm_state_size(const nmx_norm* norm, int64_t* n_bytes);
int nmx_norm_state_export(nmx_norm* norm, void* dst, int64_t n_bytes);
int nmx_norm_state_import(nmx_norm* norm, const void* src, int64_t n_bytes);

/* The feature normaliser INSIDE the launch sequence: once attached, every nmx_process_batch /
 * nmx_process_window of `plan` normalises its output rows on the device (same stream, hop order, chunk by
 * chunk) before they are copied back -- the features never make the extra host round trip of a separate
 * nmx_norm_process call.  norm == NULL detaches.  The normaliser must live on the plan's device and have
 * n_cols == the plan's n_outputs; the plan does not own it. */
int nmx_plan_attach_norm(nmx_plan* plan, nmx_norm* norm);

/* Page-locked host memory for the buffers handed to memspace-0 calls: copies from / to it run at the full
 * PCIe rate and truly asynchronously (pageable memory is staged by the runtime at a fraction of that and
 * blocks the calling thread).  Plain pointers; free with nmx_host_free. */
int nmx_host_alloc(int64_t n_bytes, void** out);
int nmx_host_free(void* p);

/* Device memory of destroyed plans is recycled by later plans (a fresh reference-style DataProcessor per run costs no
 * hipMalloc / hipFree round trips): bounded by NMX_DEVICE_POOL_MB (default 8192, 0: off) and by an eighth of the device's
 * free memory, aged out after NMX_POOL_KEEP_PLANS (4) destroyed plans without reuse.  This call hands idle blocks back to
 * the driver NOW, until at most keep_bytes stay cached (0: all) -- for a process that shares its GPU (a second rank,
 * another allocator) or is done with the engine for a while.  *freed (may be NULL) = bytes released.  Thread safe. */
int nmx_device_pool_trim(int64_t keep_bytes, int64_t* freed);

/* The stand-alone ReReferencer in FLOAT64 (processing/rereference.py:88-102: `ref_matrix @ data` on the float64 array the
 * reference holds; its tests compare at rtol 1e-7, tests/test_rereference.py:57-182): y[n_out][ldy] = R[n_out][n_in] x
 * x[n_in][ldx] over n_samples columns of any number, float64 in, float64 accumulation in the order of the columns of R,
 * float64 out -- IEEE semantics of a dense product (a NaN / inf sample reaches every row, as 0 x NaN does there).  Host
 * pointers; one kernel on `device`, HBM bound.  Inside a plan the re-reference runs on the fp32 windows (1e-5). */
int nmx_reref_f64(int device, const double* ref_matrix, int n_out, int n_in, const double* x, int64_t ldx,
                  int64_t n_samples, double* y, int64_t ldy);

/* The stand-alone Resampler in FLOAT64 (processing/resample.py:42-60: mne.filter.resample(x.astype(float64), up = ratio,
 * down = 1) -- FFT method, boxcar window, npad "auto", reflect_limited padding) for ANY window length (the reference's tests
 * resample 10 s at 4 kHz in one call, tests/test_nm_resample.py:8-47): y[n_channels][ldy] <- x[n_channels][ldx], n_out =
 * round(ratio * n_samples) (round-half-even, as Python's) samples per row.  Host pointers; power-of-two Stockham transforms
 * over HBM in float64, Bluestein's chirp convolution for a resampled padded length that is not a power of two
 * (nmx_k_resample64.h).  NaN / inf samples spread over their row as in the reference.  Inside a plan the resampler runs on
 * fp32 windows in LDS (raw_window / resample_ratio of the plan description). */
int nmx_resample_f64(int device, const double* x, int64_t ldx, int n_channels, int64_t n_samples, double ratio, double* y,
                     int64_t ldy, int64_t n_out);

/* Host-side staging passes of the boundary (no device work; a few threads of their own, n_threads <= 0: an eighth of the machine, 4 .. 16).  They
 * replace what a NumPy host does at 1 - 3 GB/s around a batch call -- the reference hands float64 rows
 * (stream/stream.py:298-310) and expects a float64 table in its own column order (stream/stream.py:319-343):
 *   nmx_host_stage_rows   dst[r][t] = (float)(src[rows[r]][t] - sub[rows[r]])  for r < n_rows, t in [t0, t1)
 *                         src float32 or float64 (src_is_f64), rows == NULL: rows 0 .. n_rows - 1, sub (per SOURCE row,
 *                         float64, subtracted before the cast: the offset split of nmx_plan_set_offsets) may be NULL;
 *   nmx_host_group_sums   sum[t] = sum over r of nan_to_num((float)src[rows[r]][t]) in float64, rows in order, t in
 *                         [t0, t1): the channel sum of a re-reference group (processing/rereference.py:52-100) for the
 *                         parts of a multi-device stream, which each hold some of the group's rows;
 *   nmx_host_stage_parts  both of the above in ONE pass over src rows 0 .. n_src_rows - 1, samples [t0, t1), block by
 *                         block: row j goes to dst_rows[j] (pointer to sample 0 of its destination row; NULL: nowhere),
 *                         group g = source rows group_rows[group_ptr[g] .. group_ptr[g + 1]) is summed into sum_rows[g];
 *   nmx_host_widen_rows   dst[r][runs[k][0] + i] = (double)src[r][runs[k][1] + i],  i < runs[k][2], r in [r0, r1): the
 *                         float32 rows of one part into the float64 table, runs[n_runs][3] = (first column in dst, first
 *                         column in src, length). */
int nmx_host_stage_rows(float* dst, int64_t ld_dst, const void* src, int src_is_f64, int64_t ld_src, const int32_t* rows,
                        int32_t n_rows, int64_t t0, int64_t t1, const double* sub, int32_t n_threads);
int nmx_host_group_sums(double* sum, const void* src, int src_is_f64, int64_t ld_src, const int32_t* rows, int32_t n_rows,
                        int64_t t0, int64_t t1, int32_t n_threads);
int nmx_host_stage_parts(const void* src, int src_is_f64, int64_t ld_src, int32_t n_src_rows, int64_t t0, int64_t t1,
                         float* const* dst_rows, int32_t n_groups, const int32_t* group_ptr, const int32_t* group_rows,
                         double* const* sum_rows, int32_t n_threads);
int nmx_host_widen_rows(double* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t r0, int64_t r1,
                        const int64_t* runs, int32_t n_runs, int32_t n_threads);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NMX_H */

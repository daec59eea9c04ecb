"""NMPreprocessor-compatible pre-processing + host-side feature normalisation.

  NotchFilter   filter/notch_filter.py:9-93     process(data[C, W] or [W]) -> data   (HIP FIR kernel; any length)
  ReReferencer  processing/rereference.py:9-102 process(data) = ref_matrix @ data in float64 (HIP kernel nmx_reref_f64)
  PreprocessingFilter processing/filter_preprocessing.py:44-94  chained single FIRs (HIP FIR kernel; any length)
  RawNormalizer processing/normalization.py:113-116  mean / zscore / median / zscore-median / robust / minmax / quantile /
                power of the raw window against its history
  Resampler     processing/resample.py:19-60    FFT resampling in float64, any length (HIP kernels nmx_resample_f64;
                restated MNE algorithm, parity unpinned); identity at ratio 1
  FeatureNormalizer processing/normalization.py:119-122 -- the reference's one-vector call shape over the device
                normaliser below.
  DeviceFeatureNormalizer  the same post-processing for every method (mean, median, zscore (default), zscore-median,
                robust, minmax, quantile, power) as a HIP scan over a whole batch of hops (libnmx nmx_norm_*,
                SURVEY 8f "next" #1).

These are the STAND-ALONE objects (what a user of the reference constructs by hand, and what its tests exercise); inside
``DataProcessor`` / ``Stream`` the same stages are fused into the plan and run on its fp32 windows.
"""

from __future__ import annotations

import numpy as np

from . import channels as chmod
from . import fir_design
from .engine import MAX_PLAN_WINDOW, HotPathEngine, long_segments
from .settings import NMSettings


def _pre_engine(C, W, sfreq, notch_taps=None, pre_taps=None, raw_norm=None):
    """A plan that only pre-processes (feature "return_raw") ``C`` channels x ``W`` samples: the stand-alone FIR stages
    and the raw normaliser reuse the kernels of the fused path."""
    s = NMSettings.get_default()
    names = [f"c{i}" for i in range(C)]
    if raw_norm is not None:
        return HotPathEngine(s, names, sfreq, features=["return_raw"], raw_norm=raw_norm, window=W)
    if pre_taps is not None:
        return HotPathEngine(s, names, sfreq, features=["return_raw"], pre_taps=pre_taps, window=W)
    return HotPathEngine(s, names, sfreq, features=["return_raw"], notch_taps=notch_taps, window=W)


def _windowed(engines: dict, make, data: np.ndarray, halo: int) -> np.ndarray:
    """``process`` of a stand-alone FIR pre-processor: one window (of whatever length; 1-D like MNE's filters take it),
    or a recording longer than a plan's window in exact segments (engine.long_segments)."""
    data = np.asarray(data, np.float64)
    x = data[None] if data.ndim == 1 else data
    C_, T = x.shape

    def eng(W):
        if (C_, W) not in engines:
            engines[(C_, W)] = make(C_, W)
        return engines[(C_, W)]

    if T <= MAX_PLAN_WINDOW:
        y = eng(T).preprocess_window(x)
    else:
        e = eng(MAX_PLAN_WINDOW)
        y = np.empty((e.C, T), np.float64)
        for lo, a, b in long_segments(T, halo):
            y[:, a:b] = e.preprocess_window(x[:, lo:lo + MAX_PLAN_WINDOW])[:, a - lo:b - lo]
    return y[0] if data.ndim == 1 else y


class NotchFilter:
    def __init__(self, sfreq: float, line_noise: float | None = None, freqs=None,
                 notch_widths=3, trans_bandwidth: float = 6.8) -> None:
        if line_noise is None and freqs is None:
            raise ValueError("Either line_noise or freqs must be defined if notch_filter is activated.")
        self.sfreq = sfreq
        self.filter_bank = fir_design.notch_bank(sfreq, line_noise, notch_widths, trans_bandwidth, freqs=freqs)
        self._engines: dict = {}

    def process(self, data: np.ndarray) -> np.ndarray:
        if self.filter_bank is None:
            return data
        return _windowed(self._engines, lambda C_, W: _pre_engine(C_, W, self.sfreq, notch_taps=self.filter_bank),
                         data, (len(self.filter_bank) - 1) // 2)


class ReReferencer:
    """processing/rereference.py:9-102 as a stand-alone object: ``ref_matrix @ data`` in FLOAT64 on the device
    (nmx_reref_f64: the reference's own tests compare the result with float64 arithmetic at rtol 1e-7), any number of
    samples.  Inside ``DataProcessor`` / ``Stream`` the re-reference is a stage of the plan on its fp32 windows."""

    def __init__(self, sfreq: float, channels, device: int = 0) -> None:
        self.sfreq = sfreq
        self.ref_matrix = chmod.reref_matrix(chmod.load_channels(channels))
        self.device = int(device)

    def process(self, data: np.ndarray) -> np.ndarray:
        if self.ref_matrix is None:
            return data
        from ._lib import get_library

        lib = get_library()
        R = np.ascontiguousarray(self.ref_matrix, dtype=np.float64)
        x = np.asarray(data, dtype=np.float64)
        if x.ndim != 2 or x.shape[0] != R.shape[1]:
            raise ValueError(f"expected data with {R.shape[1]} rows, got {x.shape}")
        if x.strides[1] != 8 or x.strides[0] < 8 * x.shape[1] or x.strides[0] % 8:
            x = np.ascontiguousarray(x)
        y = np.empty((R.shape[0], x.shape[1]), np.float64)
        lib.check(lib.lib.nmx_reref_f64(self.device, R.ctypes.data, R.shape[0], R.shape[1], x.ctypes.data,
                                        x.strides[0] // 8 if x.shape[0] > 1 else max(x.shape[1], 1), x.shape[1],
                                        y.ctypes.data, max(y.shape[1], 1)))
        return y


class PreprocessingFilter:
    """processing/filter_preprocessing.py:44-94: the enabled band / low- / high-pass FIRs applied one
    after the other to the window (zero-padded "same" convolutions) on the device."""

    def __init__(self, settings, sfreq: float) -> None:
        self.sfreq = float(sfreq)
        self.taps = fir_design.preprocessing_filter_bank(settings.preprocessing_filter, self.sfreq)
        self._engines: dict = {}

    def process(self, data: np.ndarray) -> np.ndarray:
        if not self.taps:
            return data
        return _windowed(self._engines, lambda C_, W: _pre_engine(C_, W, self.sfreq, pre_taps=self.taps),
                         data, sum((len(t) - 1) // 2 for t in self.taps))


class RawNormalizer:
    """processing/normalization.py:113-116 (Normalizer with type "raw") on the device for "mean", "zscore",
    "median", "zscore-median" and the scikit-learn based "robust" / "minmax" / "quantile" / "power"
    (nmx_k_rawnorm.h: sliding sums; for the order statistics a sorted copy of the history merged once per hop, from
    which "quantile" fits its 300-entry table -- histories of more than 10 000 samples through a uniformly random
    10 000-subset like scikit-learn's; "power" fits Yeo-Johnson's lambda by the bounded Brent search of
    nmx_k_power.h); stateful like the reference (one instance = one stream)."""

    def __init__(self, sfreq: float, settings, **kwargs) -> None:
        rs = settings.raw_normalization_settings
        self.sfreq = float(sfreq)
        self._spec = (rs.normalization_method, rs.clip, int(rs.normalization_time_s * sfreq),
                      int(sfreq / settings.sampling_rate_features_hz))
        if rs.normalization_method not in ("mean", "zscore", "median", "zscore-median", "robust", "minmax",
                                           "quantile", "power"):
            raise ValueError(f"unknown raw_normalization method {rs.normalization_method!r}")
        self._engine = None

    def process(self, data: np.ndarray) -> np.ndarray:
        data = np.asarray(data, dtype=np.float64)
        if self._engine is None:
            self._engine = _pre_engine(data.shape[0], data.shape[1], self.sfreq, raw_norm=self._spec)
        return self._engine.preprocess_window(data)


class Resampler:
    """processing/resample.py:19-60 as a stand-alone object: ``mne.filter.resample(data.astype(float64), up = new / old,
    down = 1)`` along the last axis (FFT method, reflect_limited padding) in FLOAT64 on the device, for any window length
    (nmx_resample_f64; the reference's tests resample 10 s of data in one call); identity at ratio 1 like the reference.
    Inside ``DataProcessor`` / ``Stream`` the resampler is a stage of the plan on its fp32 windows."""

    def __init__(self, sfreq: float, resample_freq_hz: float, device: int = 0, **kwargs) -> None:
        self.sfreq = float(sfreq)
        self.resample_freq_hz = float(resample_freq_hz)
        ratio = float(resample_freq_hz / sfreq)
        self.up = 0.0 if ratio == 1.0 else ratio
        self.device = int(device)

    def process(self, data: np.ndarray) -> np.ndarray:
        if not self.up:
            return data
        from ._lib import get_library

        lib = get_library()
        data = np.asarray(data, dtype=np.float64)
        x = np.ascontiguousarray(data.reshape(-1, data.shape[-1]))
        n_out = int(round(self.up * x.shape[1]))
        y = np.empty((x.shape[0], n_out), np.float64)
        if x.shape[0] and x.shape[1]:
            lib.check(lib.lib.nmx_resample_f64(self.device, x.ctypes.data, x.shape[1], x.shape[0], x.shape[1], self.up,
                                               y.ctypes.data, max(n_out, 1), n_out))
        return y.reshape(data.shape[:-1] + (n_out,))


class FeatureNormalizer:
    """processing/normalization.py:119-122, the reference's call shape (one feature vector per call) ON THE DEVICE:
    a ``DeviceFeatureNormalizer`` created at the first call, when the vector length is known.  Every method of
    normalization.py:57-70 runs there ("power" included: nmx_k_power.h); there is no host implementation, and an
    unknown method name raises ``NotImplementedError``."""

    def __init__(self, settings) -> None:
        method = settings.feature_normalization_settings.normalization_method
        if method not in DeviceFeatureNormalizer.METHODS:
            raise NotImplementedError(
                f"feature normalization_method {method!r} has no device implementation (available: "
                f"{', '.join(DeviceFeatureNormalizer.METHODS)})")
        self._settings = settings
        self._dev = None

    def process(self, data: np.ndarray) -> np.ndarray:
        data = np.asarray(data)
        if self._dev is None:
            self._dev = DeviceFeatureNormalizer(self._settings, int(data.shape[-1]))
        return self._dev.process(data)


class DeviceFeatureNormalizer:
    """processing/normalization.py:31-111 on the GPU for normalization_method in {"mean", "median", "zscore",
    "zscore-median"} and the scikit-learn based "robust", "minmax", "quantile" and "power" (RobustScaler / MinMaxScaler /
    QuantileTransformer(n_quantiles=300) / PowerTransformer fitted on nan_to_num(history) every hop: restated in
    nmx_k_norm.h and nmx_k_power.h; histories are at most N = normalization_time_s * sampling_rate_features_hz rows, far
    below QuantileTransformer's random subsampling threshold of 10 000).

    ``process(row)`` keeps the reference's call shape (one feature vector per hop);
    ``process_batch(rows)`` normalises ``rows[n_hops, n_features]`` with the same hop-by-hop
    semantics in one kernel launch (rows may also be a device pointer, see ``process_device``).
    """

    METHODS = {"mean": 0, "zscore": 1, "median": 2, "zscore-median": 3, "robust": 4, "minmax": 5, "quantile": 6,
               "power": 7}

    def __init__(self, settings, n_features: int, colmask=None, device: int = 0, lib=None) -> None:
        import ctypes as C

        from ._lib import get_library

        s = settings.feature_normalization_settings
        if s.normalization_method not in self.METHODS:
            raise ValueError(f"normalization_method {s.normalization_method!r} has no device implementation")
        self.method = s.normalization_method
        self.clip = float(s.clip) if s.clip else 0.0
        self.num_samples_normalize = int(s.normalization_time_s * settings.sampling_rate_features_hz)
        self.n_features = int(n_features)
        self._lib = lib if lib is not None else get_library()
        self._h = C.c_void_p()
        mask = None
        if colmask is not None:
            mask = np.ascontiguousarray(colmask, dtype=np.uint8)
            if mask.shape != (self.n_features,):
                raise ValueError("colmask must have one entry per feature")
        self._lib.check(self._lib.lib.nmx_norm_create(
            int(device), self.n_features, self.METHODS[self.method], self.clip, self.num_samples_normalize,
            mask.ctypes.data if mask is not None else None, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.lib.nmx_norm_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def process_batch(self, rows: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.n_features).copy()
        self._lib.check(self._lib.lib.nmx_norm_process(self._h, out.ctypes.data, self.n_features,
                                                       out.shape[0], 0, None))
        return out

    def process_device(self, ptr: int, ld: int, n_rows: int, stream: int | None = None) -> None:
        """In place on device memory ``float32[n_rows][ld]`` (asynchronous on ``stream``)."""
        self._lib.check(self._lib.lib.nmx_norm_process(self._h, ptr, int(ld), int(n_rows), 1, stream))

    def process(self, data: np.ndarray) -> np.ndarray:
        return self.process_batch(np.asarray(data)[None])[0].astype(np.float64)

    def reset(self) -> None:
        self._lib.check(self._lib.lib.nmx_norm_reset(self._h))

    def export_state(self) -> bytes:
        import ctypes as C

        n = C.c_int64()
        self._lib.check(self._lib.lib.nmx_norm_state_size(self._h, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._lib.check(self._lib.lib.nmx_norm_state_export(self._h, buf, n.value))
        return buf.raw

    def import_state(self, state: bytes) -> None:
        self._lib.check(self._lib.lib.nmx_norm_state_import(self._h, state, len(state)))

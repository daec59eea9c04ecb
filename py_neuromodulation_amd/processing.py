"""NMPreprocessor-compatible pre-processing + host-side feature normalisation.

  NotchFilter   filter/notch_filter.py:9-93     process(data[C, W]) -> data   (HIP FIR kernel)
  ReReferencer  processing/rereference.py:9-102 process(data) = ref_matrix @ data  (HIP kernel)
  Resampler     processing/resample.py:19-60    identity at ratio 1 (all BASELINE configs);
                                                other ratios: NotImplementedError (MNE parity unpinned)
  FeatureNormalizer processing/normalization.py:31-111 -- post-processing of the tiny per-hop
                feature vector, sequential over hops; stays on the host (SURVEY 8f "next" #1).
"""

from __future__ import annotations

import numpy as np

from . import channels as chmod
from . import fir_design
from .engine import HotPathEngine
from .settings import NMSettings


def _pre_engine(C_in, W, sfreq, notch_taps=None, ref_matrix=None, C_out=None):
    s = NMSettings.get_default()
    C = C_out if C_out is not None else C_in
    return HotPathEngine(s, [f"c{i}" for i in range(C)], sfreq, features=["return_raw"],
                         notch_taps=notch_taps, ref_matrix=ref_matrix, window=W)


class NotchFilter:
    def __init__(self, sfreq: float, line_noise: float | None = None, freqs=None,
                 notch_widths=3, trans_bandwidth: float = 6.8) -> None:
        if line_noise is None and freqs is None:
            raise ValueError("Either line_noise or freqs must be defined if notch_filter is activated.")
        if freqs is not None:
            raise NotImplementedError("explicit notch `freqs` are not supported; pass line_noise")
        self.sfreq = sfreq
        self.filter_bank = fir_design.notch_bank(sfreq, line_noise, float(np.atleast_1d(notch_widths)[0]),
                                                 trans_bandwidth)
        self._engines: dict = {}

    def process(self, data: np.ndarray) -> np.ndarray:
        if self.filter_bank is None:
            return data
        data = np.asarray(data, np.float64)
        if data.shape not in self._engines:
            eng = _pre_engine(data.shape[0], data.shape[1], self.sfreq, notch_taps=self.filter_bank)
            self._engines[data.shape] = eng
        return self._engines[data.shape].preprocess_window(data)


class ReReferencer:
    def __init__(self, sfreq: float, channels) -> None:
        self.sfreq = sfreq
        self.ref_matrix = chmod.reref_matrix(chmod.load_channels(channels))
        self._engines: dict = {}

    def process(self, data: np.ndarray) -> np.ndarray:
        if self.ref_matrix is None:
            return data
        data = np.asarray(data, np.float64)
        if data.shape not in self._engines:
            self._engines[data.shape] = _pre_engine(data.shape[0], data.shape[1], self.sfreq,
                                                    ref_matrix=self.ref_matrix,
                                                    C_out=self.ref_matrix.shape[0])
        return self._engines[data.shape].preprocess_window(data)


class Resampler:
    def __init__(self, sfreq: float, resample_freq_hz: float, **kwargs) -> None:
        ratio = float(resample_freq_hz / sfreq)
        self.up = 0.0 if ratio == 1.0 else ratio

    def process(self, data: np.ndarray) -> np.ndarray:
        if not self.up:
            return data
        raise NotImplementedError(
            "raw_resampling at a ratio != 1 is outside the accelerated path (MNE's resampler is "
            "parity-unpinned, SURVEY.md 8c); set raw_resampling_settings.resample_freq_hz == sfreq")


class FeatureNormalizer:
    """processing/normalization.py:31-111 for the NumPy methods and the sklearn ones."""

    def __init__(self, settings) -> None:
        s = settings.feature_normalization_settings
        self.method = s.normalization_method
        self.clip = s.clip
        self.num_samples_normalize = int(s.normalization_time_s * settings.sampling_rate_features_hz)
        self.previous = np.empty((0, 0))
        self._sk = None
        if self.method in ("quantile", "power", "robust", "minmax"):
            import sklearn.preprocessing as skpp

            self._sk = {"quantile": lambda: skpp.QuantileTransformer(n_quantiles=300),
                        "robust": skpp.RobustScaler, "minmax": skpp.MinMaxScaler,
                        "power": skpp.PowerTransformer}[self.method]()

    def process(self, data: np.ndarray) -> np.ndarray:
        if self.previous.size == 0:
            self.previous = data
            return data
        self.previous = np.vstack((self.previous, data))
        prev = self.previous
        with np.errstate(divide="ignore", invalid="ignore"):
            if self._sk is not None:
                out = self._sk.fit(np.nan_to_num(prev)).transform(data[None]).squeeze()
            else:
                has_nan = bool(np.any(np.isnan(prev.sum(axis=0))))
                mean = (np.nanmean if has_nan else np.mean)(prev, axis=0)
                if self.method == "mean":
                    out = (data - mean) / mean
                elif self.method == "median":
                    med = (np.nanmedian if has_nan else np.median)(prev, axis=0)
                    out = (data - med) / med
                else:
                    std = (np.nanstd if has_nan else np.std)(prev, axis=0)
                    std[std == 0] = 1
                    if self.method == "zscore":
                        out = (data - mean) / std
                    elif self.method == "zscore-median":
                        out = (data - (np.nanmedian if has_nan else np.median)(prev, axis=0)) / std
                    else:
                        raise ValueError(f"unknown normalization_method {self.method}")
        if self.clip:
            out = out.clip(min=-self.clip, max=self.clip)
        self.previous = self.previous[-self.num_samples_normalize + 1:]
        return np.nan_to_num(out)

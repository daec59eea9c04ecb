"""NMFeature-compatible plugins backed by the HIP engine (drop-in for features/*.py).

Every class has the reference's plugin signature (utils/types.py:59-77)

    cls(settings, ch_names, sfreq).calc_feature(data[C, W] float64) -> dict[str, float]

with identical keys, key order and (to fp32 tolerance) values, so it can be registered with
``nm.add_custom_feature`` or swapped into ``py_neuromodulation.features``.
``HotPathFeatures`` is the fused variant: all enabled features of a settings object from ONE
device round-trip -- what ``DataProcessor`` here uses instead of one call per feature
(features/feature_processor.py:79-84).
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np

from .engine import MAX_PLAN_WINDOW, HotPathEngine, long_segments

FEATURE_DICT = {  # features/feature_processor.py:10-25 (hot-path subset)
    "raw_hjorth": "Hjorth", "return_raw": "Raw", "bandpass_filter": "BandPower", "stft": "STFT",
    "fft": "FFT", "welch": "Welch", "sharpwave_analysis": "SharpwaveAnalyzer", "bursts": "Bursts",
    "linelength": "LineLength",
}


class _EngineFeature:
    feature_name: str = ""

    def __init__(self, settings, ch_names: Sequence[str], sfreq: float, **engine_kwargs) -> None:
        if hasattr(settings, "validate"):
            settings.validate()
        self.settings = settings
        self.ch_names = list(ch_names)
        self.sfreq = sfreq
        self._kwargs = engine_kwargs
        if not self.ch_names:
            # the reference's classes accept an empty channel list (tests/test_sharpwave.py:46-62 constructs one to
            # have the settings checked): the plan description is derived -- every settings error raised -- for one
            # stand-in channel, and there is nothing to compute.
            HotPathEngine(settings, ["_"], sfreq, features=self._feature_list(settings), dry_run=True, **engine_kwargs)
            self.engine, self.keys, self._engines = None, [], {}
            return
        self.engine = HotPathEngine(settings, self.ch_names, sfreq,
                                    features=self._feature_list(settings), **engine_kwargs)
        self.keys = self.engine.keys
        self._engines = {self.engine.W_in: self.engine}   # by window length, see `calc_feature`

    def _feature_list(self, settings):
        return [self.feature_name]

    def _engine_for(self, n_samples: int) -> HotPathEngine:
        """A sampling rate that is not a whole number of samples per segment makes the reference's generator cut
        windows of two lengths (stream/generator.py:41-53; tests/test_timing.py:43-76), and its feature classes take
        whatever length arrives: one plan per length here, and what carries over from hop to hop (burst history,
        Kalman filters) travels with the stream when the length changes (nmx_state_export / _import; the layout depends
        on the settings, not on the length) -- as `Stream.run` does for its own ragged runs."""
        eng = self._engines.get(n_samples)
        if eng is None:
            kw = dict(self._kwargs)
            kw["raw_window" if self.engine.resample_ratio else "window"] = n_samples
            eng = HotPathEngine(self.settings, self.ch_names, self.sfreq, features=self._feature_list(self.settings), **kw)
            self._engines[n_samples] = eng
        if eng is not self.engine:
            state = self.engine.export_state()
            if state:
                eng.import_state(state)
            self.engine = eng
        return eng

    def calc_feature(self, data: np.ndarray) -> dict:
        if self.engine is None:
            return {}
        n = np.shape(data)[-1]
        eng = self.engine if n == self.engine.W_in else self._engine_for(n)
        out = eng.process_window(data)
        return dict(zip(self.keys, out.tolist()))


class Hjorth(_EngineFeature):
    """features/hjorth_raw.py:18-42"""
    feature_name = "raw_hjorth"


class Raw(_EngineFeature):
    """features/hjorth_raw.py:45-57"""
    feature_name = "return_raw"


class LineLength(_EngineFeature):
    """features/linelength.py:11-21"""
    feature_name = "linelength"


class FFT(_EngineFeature):
    """features/oscillatory.py:58-119"""
    feature_name = "fft"


class Welch(_EngineFeature):
    """features/oscillatory.py:122-182"""
    feature_name = "welch"


class STFT(_EngineFeature):
    """features/oscillatory.py:185-250"""
    feature_name = "stft"


class BandPower(_EngineFeature):
    """features/bandpower.py:98-207 (FIR bank of filter/mne_filter.py fused with the tail statistics)"""
    feature_name = "bandpass_filter"


class Bursts(_EngineFeature):
    """features/bursts.py:60-298; stateful (percentile ring) exactly like the reference object."""
    feature_name = "bursts"


class SharpwaveAnalyzer(_EngineFeature):
    """features/sharpwaves.py:100-465"""
    feature_name = "sharpwave_analysis"


class HotPathFeatures(_EngineFeature):
    """All enabled hot-path features of ``settings`` in reference order from one launch sequence."""

    def _feature_list(self, settings):
        return None  # engine takes settings.features.get_enabled()


class MNEFilter:
    """filter/mne_filter.py:35-128: FIR bank ``filter_data(x[C, W]) -> (C, n_bands, W)``."""

    def __init__(self, f_ranges, sfreq, filter_length="999ms", l_trans_bandwidth=4,
                 h_trans_bandwidth=4, verbose=None) -> None:
        from . import fir_design

        if isinstance(filter_length, str):
            low = filter_length.lower()
            mult = 1e-3 if low.endswith("ms") else 1.0
            filter_length = int(np.ceil(float(low.rstrip("ms")) * mult * sfreq))
        self.filter_bank = fir_design.band_pass_bank(f_ranges, sfreq, filter_length,
                                                     l_trans_bandwidth, h_trans_bandwidth)
        self.num_filters = len(self.filter_bank)
        self.sfreq = sfreq
        self._engines: dict = {}

    def _engine(self, C_: int, W: int) -> HotPathEngine:
        from .settings import NMSettings

        key = (C_, W)
        if key not in self._engines:
            s = NMSettings.get_default()
            s.frequency_ranges_hz = {f"b{i}": [1, 2] for i in range(self.num_filters)}
            s.bandpass_filter_settings.segment_lengths_ms = {f"b{i}": 1 for i in range(self.num_filters)}
            self._engines[key] = HotPathEngine(s.validate(), [f"c{i}" for i in range(C_)], self.sfreq,
                                               features=["bandpass_filter"], bank_taps=self.filter_bank, window=W)
        return self._engines[key]

    def filter_data(self, data: np.ndarray) -> np.ndarray:
        data = np.asarray(data, np.float64)
        if data.ndim > 2:
            raise ValueError(f"Data must have one or two dimensions. Got: {data.ndim} dimensions.")
        if data.ndim == 1:
            data = data[None]
        C_, T = data.shape
        if T <= MAX_PLAN_WINDOW:
            return self._engine(C_, T).filter_window(data)
        # A recording longer than one plan's window (filter/mne_filter.py:100-128 convolves any length, "same"
        # alignment, zeros outside; tests/test_nm_filter.py filters 10 s at 4 kHz): see `long_segments`.
        eng = self._engine(C_, MAX_PLAN_WINDOW)
        out = np.empty((C_, self.num_filters, T), np.float64)
        for lo, a, b in long_segments(T, max((len(t) - 1) // 2 for t in self.filter_bank)):
            out[:, :, a:b] = eng.filter_window(data[:, lo:lo + MAX_PLAN_WINDOW])[:, :, a - lo:b - lo]
        return out

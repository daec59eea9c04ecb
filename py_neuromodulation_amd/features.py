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

from .engine import HotPathEngine

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
        self.engine = HotPathEngine(settings, self.ch_names, sfreq,
                                    features=self._feature_list(settings), **engine_kwargs)
        self.keys = self.engine.keys

    def _feature_list(self, settings):
        return [self.feature_name]

    def calc_feature(self, data: np.ndarray) -> dict:
        out = self.engine.process_window(data)
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

    def filter_data(self, data: np.ndarray) -> np.ndarray:
        from .settings import NMSettings

        data = np.asarray(data, np.float64)
        if data.ndim > 2:
            raise ValueError(f"Data must have one or two dimensions. Got: {data.ndim} dimensions.")
        if data.ndim == 1:
            data = data[None]
        key = data.shape
        if key not in self._engines:
            s = NMSettings.get_default()
            s.frequency_ranges_hz = {f"b{i}": [1, 2] for i in range(self.num_filters)}
            s.bandpass_filter_settings.segment_lengths_ms = {f"b{i}": 1 for i in range(self.num_filters)}
            eng = HotPathEngine(s.validate(), [f"c{i}" for i in range(data.shape[0])], self.sfreq,
                                features=["bandpass_filter"], bank_taps=self.filter_bank,
                                window=data.shape[1])
            self._engines[key] = eng
        return self._engines[key].filter_window(data)

"""Settings surface of the hot path: same field names as the reference's ``NMSettings``.

The engine reads settings duck-typed (attribute / ``[...]`` access), so either the
reference's pydantic ``NMSettings`` object or this light-weight loader of the same YAML /
JSON schema can be handed to ``Stream`` / ``DataProcessor`` / the feature plugins.
Only the fields the hot path consumes are modelled; unknown
fields are kept as plain attributes so a reference settings file loads unchanged.

Reference surface mirrored here:
  stream/settings.py:41-69   FeatureSelector, PostprocessingSettings, DEFAULT_PREPROCESSORS
  stream/settings.py:72-124  NMSettings fields
  stream/settings.py:152-201 validate_settings (band names, >=1 feature, band-pass segments)
  stream/settings.py:203-299 reset / set_fast_compute / load / from_file / get_default
  utils/types.py:83-162      FrequencyRange, BoolSelector
"""

from __future__ import annotations

import copy
import json
import math
import weakref
from pathlib import Path
from typing import Any

import numpy as np

__all__ = ["NMSettings", "FrequencyRange", "BoolSelector", "SettingsError", "validation_error"]


class SettingsError(ValueError):
    """Raised for invalid settings (the reference raises pydantic's ValidationError,
    which is also a ValueError subclass)."""


def validation_error(message: str, location=()) -> ValueError:
    """The error a plugin class raises for settings the reference rejects with ``create_validation_error``
    (utils/pydantic_extensions.py:26-56, e.g. features/bursts.py:68-73): pydantic's own ``ValidationError`` where
    pydantic is installed -- callers of the reference catch that type, and it cannot be subclassed -- and
    ``SettingsError`` where it is not.  Both are ``ValueError``s."""
    try:
        from pydantic_core import InitErrorDetails, ValidationError
    except ImportError:   # pragma: no cover - pydantic comes with the reference
        return SettingsError(message)
    return ValidationError.from_exception_data(
        "Validation Error",
        [InitErrorDetails(type="value_error", loc=tuple(location), input=None, ctx={"error": message})],
        input_type="python", hide_input=False)


class FrequencyRange:
    """utils/types.py:83-131: (low, high) in Hz, indexable, iterable."""

    def __init__(self, frequency_low_hz, frequency_high_hz=None) -> None:
        if frequency_high_hz is None and not isinstance(frequency_low_hz, (int, float)):
            v = frequency_low_hz
            if isinstance(v, FrequencyRange):
                frequency_low_hz, frequency_high_hz = v.as_tuple()
            elif isinstance(v, dict):
                frequency_low_hz, frequency_high_hz = v["frequency_low_hz"], v["frequency_high_hz"]
            elif len(v) == 2:
                frequency_low_hz, frequency_high_hz = v
            else:
                raise SettingsError(f"Value for FrequencyRange must be a pair, got {v}")
        self.frequency_low_hz = float(frequency_low_hz)
        self.frequency_high_hz = float(frequency_high_hz)
        lo, hi = self.frequency_low_hz, self.frequency_high_hz
        if not (math.isnan(lo) or math.isnan(hi)):
            if not (lo > 0 and hi > 0):
                raise SettingsError("Frequencies must be > 0")
            if not hi > lo:
                raise SettingsError("Frequency high must be greater than frequency low")

    def __getitem__(self, i: int) -> float:
        if i == 0:
            return self.frequency_low_hz
        if i == 1:
            return self.frequency_high_hz
        raise IndexError(f"Index {i} out of range")

    def as_tuple(self):
        return (self.frequency_low_hz, self.frequency_high_hz)

    def __iter__(self):
        return iter(self.as_tuple())

    def __repr__(self) -> str:
        return f"FrequencyRange({self.frequency_low_hz}, {self.frequency_high_hz})"

    def __eq__(self, other) -> bool:
        return tuple(self) == tuple(other)


class _Node:
    """Attribute + item access over a dict of fields; keeps insertion order."""

    def __init__(self, **fields: Any) -> None:
        for k, v in fields.items():
            setattr(self, k, v)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value) -> None:
        setattr(self, key, value)

    def __contains__(self, key) -> bool:
        return key in self.__dict__

    def keys(self):
        return self.__dict__.keys()

    def to_dict(self) -> dict:
        def conv(v):
            if isinstance(v, _Node):
                return v.to_dict()
            if isinstance(v, FrequencyRange):
                return list(v.as_tuple())
            if isinstance(v, dict):
                return {k: conv(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [conv(x) for x in v]
            if isinstance(v, np.generic):      # numpy scalars assigned by callers: builtins, like pydantic's coercion
                return v.item()
            if isinstance(v, np.ndarray):
                return [conv(x) for x in v.tolist()]
            return v

        return {k: conv(v) for k, v in self.__dict__.items()}

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self.to_dict()})"


class BoolSelector(_Node):
    """utils/types.py:134-162: ordered set of boolean switches."""

    def get_enabled(self) -> list[str]:
        return [k for k, v in self.__dict__.items() if isinstance(v, bool) and v]

    def enable_all(self) -> None:
        for k, v in self.__dict__.items():
            if isinstance(v, bool):
                setattr(self, k, True)

    def disable_all(self) -> None:
        for k, v in self.__dict__.items():
            if isinstance(v, bool):
                setattr(self, k, False)

    def __iter__(self):
        return iter(self.__dict__.keys())


# names registered through add_custom_feature (features/feature_processor.py:90-108): flags of the feature selector
# that are NOT declared fields of the reference's FeatureSelector -- its get_enabled() walks model_fields only
# (utils/types.py:135-140), so a user feature never shows up there; FeatureProcessors instantiates every registered
# user feature unconditionally (feature_processor.py:52-53)
_USER_FEATURE_FIELDS: set[str] = set()


class FeatureSelector(BoolSelector):
    """stream/settings.py:41-55 + the extra attributes add_custom_feature sets on live settings objects."""

    def get_enabled(self) -> list[str]:
        return [k for k in super().get_enabled() if k not in _USER_FEATURE_FIELDS]


# ---- defaults of the fields the hot path reads (values: default_settings.yaml) ----------

_FEATURES = ["raw_hjorth", "return_raw", "bandpass_filter", "stft", "fft", "welch",
             "sharpwave_analysis", "fooof", "nolds", "coherence", "bursts", "linelength",
             "mne_connectivity", "bispectrum"]
_FEATURES_ON = {"raw_hjorth", "return_raw", "fft", "welch", "sharpwave_analysis", "bursts",
                "linelength"}
_PREPROCESSORS = ["preprocessing_filter", "notch_filter", "raw_resampling", "re_referencing",
                  "raw_normalization"]
DEFAULT_PREPROCESSORS = ["raw_resampling", "notch_filter", "re_referencing"]
_NORM_METHODS = ["mean", "median", "zscore", "zscore-median", "quantile", "power", "robust",
                 "minmax"]
_SW_FEATURES = ["peak_left", "peak_right", "num_peaks", "trough", "width", "prominence",
                "interval", "decay_time", "rise_time", "sharpness", "rise_steepness",
                "decay_steepness", "slope_ratio"]
_SW_ON = {"prominence", "interval", "sharpness"}


def _osc(window_ms: int) -> dict:
    return {"windowlength_ms": window_ms, "log_transform": True,
            "features": {"mean": True, "median": False, "std": False, "max": False},
            "return_spectrum": False}


def _default_dict() -> dict:
    return {
        "sampling_rate_features_hz": 10,
        "segment_length_features_ms": 1000,
        "frequency_ranges_hz": {"theta": [4, 8], "alpha": [8, 12], "low_beta": [13, 20],
                                "high_beta": [20, 35]},
        "features": {f: (f in _FEATURES_ON) for f in _FEATURES},
        "preprocessing": list(DEFAULT_PREPROCESSORS),
        "raw_resampling_settings": {"resample_freq_hz": 1000},
        "raw_normalization_settings": {"normalization_time_s": 30,
                                       "normalization_method": "zscore", "clip": 3},
        "preprocessing_filter": {"bandstop_filter": True, "bandpass_filter": True,
                                 "lowpass_filter": True, "highpass_filter": True,
                                 "bandstop_filter_settings": [100, 160],
                                 "bandpass_filter_settings": [3, 200],
                                 "lowpass_filter_cutoff_hz": 200, "highpass_filter_cutoff_hz": 3},
        "postprocessing": {"feature_normalization": True, "project_cortex": False,
                           "project_subcortex": False},
        "feature_normalization_settings": {"normalization_time_s": 30,
                                           "normalization_method": "zscore",
                                           "normalize_psd": False, "clip": 3},
        "fft_settings": _osc(1000),
        "welch_settings": _osc(1000),
        "stft_settings": _osc(500),
        "bandpass_filter_settings": {
            "segment_lengths_ms": {"theta": 1000, "alpha": 500, "low_beta": 333,
                                   "high_beta": 333, "low_gamma": 100, "high_gamma": 100,
                                   "HFA": 100},
            "bandpower_features": {"activity": True, "mobility": False, "complexity": False},
            "log_transform": True, "kalman_filter": False},
        "kalman_filter_settings": {"Tp": 0.1, "sigma_w": 0.7, "sigma_v": 1.0,
                                   "frequency_bands": ["theta", "alpha", "low_beta", "high_beta",
                                                       "low_gamma", "high_gamma", "HFA"]},
        "bursts_settings": {"threshold": 75, "time_duration_s": 30,
                            "frequency_bands": ["low_beta", "high_beta"],
                            "burst_features": {"duration": True, "amplitude": True,
                                               "burst_rate_per_s": True, "in_burst": True}},
        "sharpwave_analysis_settings": {
            "sharpwave_features": {f: (f in _SW_ON) for f in _SW_FEATURES},
            "filter_ranges_hz": [[5, 80], [5, 30]],
            "detect_troughs": {"estimate": True, "distance_troughs_ms": 10,
                               "distance_peaks_ms": 5},
            "detect_peaks": {"estimate": True, "distance_troughs_ms": 5,
                             "distance_peaks_ms": 10},
            "estimator": {"mean": ["interval"], "median": [], "max": ["prominence", "sharpness"],
                          "min": [], "var": []},
            "apply_estimator_between_peaks_and_troughs": True},
    }


_SELECTOR_KEYS = {"features", "postprocessing", "preprocessing_filter", "bandpower_features", "burst_features",
                  "sharpwave_features"}


def _merge(base: dict, over: dict) -> dict:
    out = copy.deepcopy(base)
    for k, v in over.items():
        # dict-valued *tables* (band tables, segment lengths) are replaced, not merged
        if (isinstance(v, dict) and isinstance(out.get(k), dict)
                and k not in ("frequency_ranges_hz", "segment_lengths_ms")):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _build(key: str, v: Any) -> Any:
    if key == "frequency_ranges_hz":
        return {str(k).replace(" ", "_"): FrequencyRange(x) for k, x in v.items()}
    if key in ("bandstop_filter_settings", "bandpass_filter_settings") and (
            isinstance(v, (list, tuple, FrequencyRange)) or (isinstance(v, dict) and "frequency_low_hz" in v)):
        return FrequencyRange(v)   # preprocessing_filter ranges (the BandPower settings dict shares a name)
    if key == "filter_ranges_hz":
        return [FrequencyRange(x) for x in v]
    if key == "segment_lengths_ms":
        return {str(k).replace(" ", "_"): int(x) for k, x in v.items()}
    if key == "features" and isinstance(v, dict) and set(v) <= {"mean", "median", "std", "max"}:
        return BoolSelector(**{k: bool(v.get(k, False)) for k in ("mean", "median", "std", "max")})
    if isinstance(v, dict):
        cls = FeatureSelector if key == "features" else BoolSelector if key in _SELECTOR_KEYS else _Node
        return cls(**{k: _build(k, x) for k, x in v.items()})
    return copy.deepcopy(v)


class NMSettings(_Node):
    """Hot-path subset of the reference's ``NMSettings`` (stream/settings.py:72-299)."""

    _instances: "weakref.WeakSet[NMSettings]"   # live objects: add_custom_feature flips the flag on each (settings.py:129-150)

    def __init__(self, **model_dict: Any) -> None:
        merged = _merge(_default_dict(), model_dict)
        super().__init__(**{k: _build(k, v) for k, v in merged.items()})
        if "frequency_bands" in self.bursts_settings:
            self.bursts_settings.frequency_bands = [
                f.replace(" ", "_") for f in self.bursts_settings.frequency_bands]
        self._check()
        for name in _USER_FEATURE_FIELDS:   # stream/settings.py:129-133
            setattr(self.features, name, True)
        NMSettings._instances.add(self)

    __hash__ = object.__hash__   # identity: the WeakSet of live instances

    # copies (copy.copy / copy.deepcopy / pickle) are live objects too: add_custom_feature / remove_custom_feature flip
    # their flags like everybody else's (the reference's class-level registry sees every instance: settings.py:129-150)
    def __copy__(self):
        new = type(self).__new__(type(self))
        new.__dict__.update(self.__dict__)
        NMSettings._instances.add(new)
        return new

    def __deepcopy__(self, memo):
        import copy

        new = type(self).__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        NMSettings._instances.add(new)
        return new

    def __setstate__(self, state):
        self.__dict__.update(state)
        NMSettings._instances.add(self)

    @classmethod
    def _add_feature(cls, feature: str) -> None:
        """stream/settings.py:140-143."""
        _USER_FEATURE_FIELDS.add(feature)
        for inst in list(cls._instances):
            setattr(inst.features, feature, True)

    @classmethod
    def _remove_feature(cls, feature: str) -> None:
        """stream/settings.py:145-148."""
        _USER_FEATURE_FIELDS.discard(feature)
        for inst in list(cls._instances):
            if feature in inst.features:
                delattr(inst.features, feature)

    # -- validation (stream/settings.py:152-201 + per-feature validators) ----------------
    def _check(self) -> None:
        errors: list[str] = []
        if not self.sampling_rate_features_hz > 0:
            errors.append("sampling_rate_features_hz must be > 0")
        if not self.segment_length_features_ms > 0:
            errors.append("segment_length_features_ms must be > 0")
        self.frequency_ranges_hz = {
            k.replace(" ", "_"): (v if isinstance(v, FrequencyRange) else FrequencyRange(v))
            for k, v in self.frequency_ranges_hz.items()}
        for p in self.preprocessing:
            if p not in _PREPROCESSORS:
                errors.append(f"Invalid preprocessing method '{p}'")
        if len(self.features.get_enabled()) == 0:
            errors.append("At least one feature must be selected.")
        for name in ("fft_settings", "welch_settings", "stft_settings"):
            s = self[name]
            if not isinstance(s.log_transform, bool):
                errors.append(f"{name}.log_transform must be a bool")
            if not (isinstance(s.windowlength_ms, int) and s.windowlength_ms > 0):
                errors.append(f"{name}.windowlength_ms must be a positive int")
        for name in ("raw_normalization_settings", "feature_normalization_settings"):
            if self[name].normalization_method not in _NORM_METHODS:
                errors.append(f"{name}.normalization_method invalid")
        bp = self.bandpass_filter_settings
        if len(bp.bandpower_features.get_enabled()) == 0:
            errors.append("Set at least one bandpower_feature to True.")
        if self.features.bandpass_filter:  # bandpower.py:51-96
            for band, seg in bp.segment_lengths_ms.items():
                if not seg <= self.segment_length_features_ms:
                    errors.append(f"segment length {seg} needs to be smaller than "
                                  f"segment_length_features_ms = {self.segment_length_features_ms}")
            for band in self.frequency_ranges_hz:
                if band not in bp.segment_lengths_ms:
                    errors.append(f"frequency range {band} needs to be defined in "
                                  "bandpass_filter_settings.segment_lengths_ms")
        sw = self.sharpwave_analysis_settings  # sharpwaves.py:88-97
        est_list = [f for e in ("mean", "median", "max", "min", "var") for f in sw.estimator[e]]
        for f in sw.sharpwave_features.get_enabled():
            if f not in est_list:
                errors.append(f"Add estimator key for {f}")
        if self.bursts_settings.threshold < 0 or self.bursts_settings.time_duration_s < 0:
            errors.append("bursts_settings threshold / time_duration_s must be >= 0")
        if errors:
            raise SettingsError("; ".join(errors))

    def validate(self) -> "NMSettings":
        """Return a validated copy (the reference's ``validate`` also copies)."""
        return NMSettings(**self.to_dict())

    # -- convenience API (stream/settings.py:203-237) -----------------------------------
    def reset(self) -> "NMSettings":
        self.features.disable_all()
        self.preprocessing = list(DEFAULT_PREPROCESSORS)
        self.postprocessing.disable_all()
        return self

    def set_fast_compute(self) -> "NMSettings":
        self.reset()
        self.features.fft = True
        self.postprocessing.feature_normalization = True
        return self

    def enable_all_features(self) -> "NMSettings":
        self.features.enable_all()
        return self

    def disable_all_features(self) -> "NMSettings":
        self.features.disable_all()
        return self

    @staticmethod
    def get_default() -> "NMSettings":
        return NMSettings()

    @staticmethod
    def get_fast_compute() -> "NMSettings":
        return NMSettings().set_fast_compute()

    @classmethod
    def load(cls, settings) -> "NMSettings":
        if isinstance(settings, cls):
            return settings.validate()
        if settings is None:
            return cls.get_default()
        if hasattr(settings, "model_dump"):   # the reference's pydantic object (stream/settings.py): the same tree
            return cls(**settings.model_dump()).validate()
        if hasattr(settings, "frequency_ranges_hz"):  # any other duck-typed settings object: read as it is
            return settings
        return cls.from_file(settings)

    @staticmethod
    def from_file(path) -> "NMSettings":
        path = Path(path)
        if path.is_dir():
            for child in sorted(path.iterdir()):
                if child.is_file() and child.suffix in (".json", ".yaml"):
                    path = child
                    break
        if path.suffix == ".json":
            with open(path) as f:
                d = json.load(f)
        elif path.suffix == ".yaml":
            import yaml

            with open(path) as f:
                d = yaml.safe_load(f)
        else:
            raise ValueError("File format not supported.")
        return NMSettings(**d)

    def to_yaml_text(self) -> str:
        import yaml

        # libyaml's emitter when PyYAML was built with it (same text, a few ms less per Stream.run)
        dumper = getattr(yaml, "CSafeDumper", None) or yaml.SafeDumper
        return yaml.dump(self.to_dict(), default_flow_style=None, Dumper=dumper)

    def save(self, out_dir=".", prefix: str = "", format: str = "yaml", text: str | None = None) -> None:
        """``text``: the serialised settings if the caller already has them (Stream.run re-uses the text of its
        previous run while the settings are unchanged)."""
        filename = f"{prefix}_SETTINGS.{format}" if prefix else f"SETTINGS.{format}"
        out = Path(out_dir) / prefix / filename
        out.parent.mkdir(parents=True, exist_ok=True)
        with open(out, "w") as f:
            if format == "json":
                json.dump(self.to_dict(), f, indent=4)
            else:
                f.write(text if text is not None else self.to_yaml_text())


NMSettings._instances = weakref.WeakSet()

"""py_neuromodulation_amd -- MI355X-native engine for py_neuromodulation's per-hop hot path
(nm.Stream -> DataProcessor.process -> filter/ -> features/), behind the reference's own
NMFeature / NMPreprocessor / DataProcessor / Stream call shapes.

Compute runs in hand-written HIP kernels (csrc/, libnmx.so) through a C ABI (include/nmx.h);
there is no CPU fallback: creating an engine without the library or without a GPU raises.
"""

import logging

from .settings import NMSettings  # noqa: F401

__version__ = "0.1.0"
logger = logging.getLogger("py_neuromodulation_amd")

# features/feature_processor.py:52-53: every DataProcessor instantiates these AFTER the built-in features and
# appends their columns (py_neuromodulation/__init__.py:60 keeps the same module-global dictionary)
user_features: dict = {}


def add_custom_feature(feature_name: str, new_feature) -> None:
    """features/feature_processor.py:90-108: register a class with the NMFeature surface
    (``__init__(settings, ch_names, sfreq)``, ``calc_feature(data[C, W]) -> dict``, utils/types.py:59-77).
    ``DataProcessor`` / ``Stream`` call it on the host with the pre-processed windows the device features
    read and append its keys, in registration order, after the built-in columns."""
    user_features[feature_name] = new_feature
    NMSettings._add_feature(feature_name)


def remove_custom_feature(feature_name: str) -> None:
    """features/feature_processor.py:111-121."""
    user_features.pop(feature_name)
    NMSettings._remove_feature(feature_name)


def __getattr__(name):  # lazy: importing the package must not require the GPU library
    if name in ("Stream",):
        from .stream import Stream

        return Stream
    if name in ("DataProcessor",):
        from .data_processor import DataProcessor

        return DataProcessor
    if name in ("HotPathEngine",):
        from .engine import HotPathEngine

        return HotPathEngine
    raise AttributeError(name)

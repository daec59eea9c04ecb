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

user_features: dict = {}


def add_custom_feature(feature_name: str, new_feature) -> None:
    """features/feature_processor.py:90-108 (host-side custom features are not run by the
    fused engine; register them on the reference's own stream instead)."""
    user_features[feature_name] = new_feature


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

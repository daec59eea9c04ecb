"""py_neuromodulation_amd -- MI355X-native engine for py_neuromodulation's per-hop hot path."""

from .settings import NMSettings  # noqa: F401

__version__ = "0.1.0"

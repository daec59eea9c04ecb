"""User-side NMFeature plugins for the tests of ``add_custom_feature`` (utils/types.py:59-77: a constructor taking
``(settings, ch_names, sfreq)`` and ``calc_feature(data[C, W]) -> dict``).  The same classes are registered on the
reference when tests/golden/make_golden.py writes ``user_features.npz`` and on this package in the parity tests.

``ChannelMean`` is the plugin of the reference's "Adding New Features" example
(examples/plot_2_example_add_feature.py:25-55): one key ``channel_mean_<channel>`` per channel holding the mean of the
window.  ``HopStats`` is stateful (a hop counter), emits a key with "psd" in its name (skipped by the feature
normaliser unless ``normalize_psd``) and keys that contain the channel name (NaN policy)."""

import numpy as np


class ChannelMean:
    def __init__(self, settings, ch_names, sfreq) -> None:
        self.ch_names = list(ch_names)

    def calc_feature(self, data: np.ndarray) -> dict:
        m = data.mean(axis=1)
        return {f"channel_mean_{ch}": m[i] for i, ch in enumerate(self.ch_names)}


class HopStats:
    def __init__(self, settings, ch_names, sfreq) -> None:
        self.ch_names = list(ch_names)
        self.sfreq = sfreq
        self.hops = 0

    def calc_feature(self, data: np.ndarray) -> dict:
        self.hops += 1
        out = {"hops_seen": float(self.hops), "window_s": data.shape[1] / self.sfreq}
        for i, ch in enumerate(self.ch_names):
            out[f"{ch}_ptp_psd_like"] = float(data[i].max() - data[i].min())
            out[f"{ch}_rms"] = float(np.sqrt(np.mean(data[i] ** 2)))
        return out

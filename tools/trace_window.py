#!/usr/bin/env python
"""A few one-window calls of one bench_window case, for `rocprofv3 --kernel-trace` (tools/rocpd_timeline.py on the result):
    python tools/trace_window.py fast1 | fast6 | headline"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def main(case):
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.data_processor import DataProcessor

    if case == "headline":
        s = bench.make_settings()
        s.postprocessing.feature_normalization = True
        data = bench.synth(256, 1000, 1000.0, 3).astype(np.float64)
    else:
        s = NMSettings.get_fast_compute()
        s.preprocessing = ["re_referencing", "notch_filter"]
        s.features.fft = True
        s.postprocessing.feature_normalization = True
        data = np.random.default_rng(0).random((1 if case == "fast1" else 6, 1000))
    dp = DataProcessor(sfreq=1000.0, settings=s, channels=chmod.get_default_channels_from_data(data), line_noise=50)
    for _ in range(40):
        dp.engine.process_window(data)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "fast1")

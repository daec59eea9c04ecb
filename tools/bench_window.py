#!/usr/bin/env python
"""The reference's REAL-TIME call shape as a measured quantity: `DataProcessor.process(window)` -> dict, one call per hop
(examples/plot_6_real_time_demo.py:54-106 times exactly this: fast-compute settings with FFT features, re-reference, notch,
z-score, on 1 and on 6 channels -- "well below 10 ms"; then the default feature set on one channel).
p50 / p99 / max over N calls of (a) the engine's one-window call alone (`HotPathEngine.process_window`: float64 window in,
float32 feature row out) and (b) `DataProcessor.process` (+ NaN policy, dict of Python floats), next to the float64 oracle's
`DataProcessor.process` on the same host core.  Also the headline workload (256 channels, all nine families).

    python tools/bench_window.py [n_calls]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def pct(v):
    v = np.sort(np.asarray(v)) * 1e3
    return {"p50_ms": round(float(v[len(v) // 2]), 4), "p99_ms": round(float(v[min(len(v) - 1, int(0.99 * len(v)))]), 4),
            "max_ms": round(float(v[-1]), 4), "calls": len(v)}


def time_calls(fn, n, warm=20):
    for _ in range(warm):
        fn()
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        out.append(time.perf_counter() - t0)
    return out


def main(n_calls=500):
    from oracle import nm_oracle as orc   # (the CPU baseline leg only)
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.data_processor import DataProcessor

    res = {}

    def fast():
        s = NMSettings.get_fast_compute()
        s.preprocessing = ["re_referencing", "notch_filter"]
        s.features.fft = True
        s.postprocessing.feature_normalization = True
        return s

    cases = [("fast_compute_fft_1ch", fast(), 1), ("fast_compute_fft_6ch", fast(), 6),
             ("default_settings_1ch", NMSettings.get_default(), 1), ("default_settings_6ch", NMSettings.get_default(), 6)]
    hs = bench.make_settings()
    hs.postprocessing.feature_normalization = True
    cases.append(("headline_all_features_256ch", hs, 256))
    rng = np.random.default_rng(0)
    for name, s, C in cases:
        data = rng.random((C, 1000)) if C <= 6 else bench.synth(C, 1000, 1000.0, 3).astype(np.float64)
        channels = chmod.get_default_channels_from_data(data)
        dp = DataProcessor(sfreq=1000.0, settings=s, channels=channels, line_noise=50)
        row = {"channels": C, "features": len(dp.keys),
               "engine_process_window": pct(time_calls(lambda: dp.engine.process_window(data), n_calls)),
               "DataProcessor_process_dict": pct(time_calls(lambda: dp.process(data), n_calls))}
        row["kernels"] = {name: dp.engine.kernels(i) for name, i in (("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6),
                                                                      ("bursts", 4), ("sharp", 5)) if dp.engine.kernels(i)}
        row["device_ms_last_call"] = round(dp.engine.timing_ms(0), 4)
        o = orc.DataProcessor(1000.0, s, channels.to_dict("list"), line_noise=50)
        n_o = 30 if C <= 6 else 6
        row["oracle_process_cpu_1core"] = pct(time_calls(lambda: o.process(data), n_o, warm=2))
        res[name] = row
        print(name, json.dumps(row), file=sys.stderr)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 500)

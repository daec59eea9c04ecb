#!/usr/bin/env python
"""Band-pass activity on 1/f ("pink") data with a large drifting offset -- the spectrum real recordings have: the
high bands carry 1e-3 of the window's power, so fp32 rounding of the FFT convolution matters most there.  Relative error
of log10 band power against the float64 oracle, for the channel-pair kernel and for the one-channel M = 2048 kernel.
    python tools/accuracy_pink.py"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from oracle import nm_oracle as orc   # checker only
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    rng = np.random.default_rng(4)
    C, W, n_hops = 16, 1000, 8
    T = W + (n_hops - 1) * 100
    f = np.fft.rfftfreq(T, 1e-3)
    spec = (rng.standard_normal((C, len(f))) + 1j * rng.standard_normal((C, len(f)))) / np.maximum(f, 0.5) ** 1.0
    x = np.fft.irfft(spec, T) * 2e4
    x += rng.uniform(-2000, 2000, (C, 1)) + np.linspace(0, 500, T)          # offset and drift
    x[1::2] *= 0.01                                                           # quiet neighbours of loud channels
    x = x.astype(np.float32)
    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = True
    s.frequency_ranges_hz = {"theta": [4, 8], "alpha": [8, 12], "low_beta": [13, 20], "high_beta": [20, 35],
                             "low_gamma": [60, 80], "high_gamma": [90, 200]}
    s.bandpass_filter_settings.segment_lengths_ms = {"theta": 1000, "alpha": 500, "low_beta": 333, "high_beta": 333,
                                                     "low_gamma": 100, "high_gamma": 100}
    s = s.validate()
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * 100
    bp = orc.BandPower(s, ch, 1000.0)
    want = np.array([[bp.calc_feature(x[:, a:a + W].astype(np.float64))[k] for k in bp.calc_feature(x[:, :W].astype(np.float64))] for a in starts])
    for flag in ("1", "0"):
        os.environ["NMX_BANK_W64C"] = flag
        eng = HotPathEngine(s, ch, 1000.0)
        got = eng.process_batch(x, starts).astype(np.float64)
        kern = eng.kernels(3)
        eng.close()
        err = np.abs(got - want)                    # log10 units
        by_band = {b: float(err[:, [i for i, k in enumerate(eng.keys) if b in k]].max()) for b in s.frequency_ranges_hz}
        print(kern, "max |d log10| =", f"{err.max():.2e}", {k: f"{v:.1e}" for k, v in by_band.items()})


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Why the STFT band means miss 1e-5 relative far more often than FFT / Welch on the headline workload -- float64
experiment, no GPU: the float64 oracle against ITSELF with one change, the pre-processed window (common average +
notch) rounded to float32 before the features (what any fp32 hand-off tensor does; relative rounding 6e-8 of samples
that carry a +-500 offset).

scipy.signal.stft(boundary="even") makes the first and the last segment SYMMETRIC about the window edge, so their
spectra are purely real (times (-1)^k): a real Gaussian amplitude has a probability DENSITY at zero (P(|X| < d) ~ d),
a complex one does not (P ~ d^2).  log10 |X| of such a near-null bin turns 1e-7-level absolute noise into 1e-4-level
errors of the band mean.  FFT / Welch bins are complex: no misses at the same noise level."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import nm_oracle as orc  # noqa: E402
from py_neuromodulation_amd import NMSettings  # noqa: E402


def main(C=32, n_hops=40):
    s = NMSettings.get_default()
    s.features.disable_all()
    for f in ("fft", "welch", "stft"):
        setattr(s.features, f, True)
    s.postprocessing.feature_normalization = False
    s.preprocessing = ["notch_filter", "re_referencing"]
    rng = np.random.default_rng(1234)
    T = 1000 + (n_hops - 1) * 100
    t = np.arange(T) / 1000.0
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + 5 * np.sin(2 * np.pi * 70 * t)
         + rng.uniform(-500, 500, (C, 1))).astype(np.float32).astype(np.float64)
    names = [f"ch{i}" for i in range(C)]
    channels = {"name": names, "rereference": ["average"] * C, "used": [1] * C, "target": [0] * C,
                "type": ["ecog"] * C, "status": ["good"] * C, "new_name": [f"{n}_avgref" for n in names]}
    dp = orc.DataProcessor(1000.0, s, channels, line_noise=50)
    feats = [orc._FEATURE_CLS[f](s, dp.ch_names_used, 1000.0) for f in ("fft", "welch", "stft")]
    rel = {"fft": [], "welch": [], "stft": []}
    for h in range(n_hops):
        w = dp.preprocess(x[:, h * 100:h * 100 + 1000])
        w32 = w.astype(np.float32).astype(np.float64)
        for name, f in zip(rel, feats):
            a, b = f.calc_feature(w), f.calc_feature(w32)
            va, vb = np.array(list(a.values())), np.array(list(b.values()))
            rel[name].append(np.abs(vb - va) / np.maximum(np.abs(va), 1e-300))
    for name, r in rel.items():
        r = np.concatenate(r)
        print(f"{name}: entries {r.size}, max rel {r.max():.2e}, p99.9 {np.quantile(r, 0.999):.2e}, "
              f"share above 1e-5 {np.mean(r > 1e-5):.4f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Soak run of the burst threshold walk: 12 000 hops (20 minutes of signal at 1 kHz, a few hundred flushes of the
one-wave kernel) against the workgroup kernel, bit for bit, with a non-stationary amplitude and a quantised channel.
    python tools/soak_walk.py [n_hops]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    n_hops = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    sfreq, C = 1000.0, 6
    W, hop = 1000, 100
    T = W + (n_hops - 1) * hop
    rng = np.random.default_rng(11)
    t = np.arange(T) / sfreq
    amp = 1 + 0.8 * np.sin(2 * np.pi * 0.01 * t) + 3 * t / t[-1] + (rng.random(T) < 1e-4) * 20
    data = (rng.standard_normal((C, T)) * 20 + 30 * amp * np.sin(2 * np.pi * 18 * t)).astype(np.float32)
    data[1] *= 1e-3
    data[2] = np.round(data[2])
    data[3] = data[3] * np.linspace(3, 0.2, T).astype(np.float32)     # shrinking power: no inserts for long stretches
    s = NMSettings.get_default().validate()
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * hop

    def run(wave):
        os.environ["NMX_THR_WAVE"] = "1" if wave else "0"
        eng = HotPathEngine(s, ch, sfreq, features=["bursts"], bank_taps=None)
        out = eng.process_batch(data, starts)
        eng.close()
        return out

    a, b = run(False), run(True)
    assert not np.isnan(a).any()
    bad = np.argwhere(a != b)
    print(f"{n_hops} hops x {C} channels: {a.size} burst features, {len(bad)} differ between the workgroup and the one-wave walk")
    assert len(bad) == 0, bad[:5]


if __name__ == "__main__":
    main()

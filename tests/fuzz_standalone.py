#!/usr/bin/env python
"""Randomised sweep of the STAND-ALONE objects (NotchFilter, PreprocessingFilter, MNEFilter on recordings of any length;
the float64 ReReferencer and Resampler) against the float64 oracle.  Not collected by pytest (test infrastructure, imports
the oracle):   python tests/fuzz_standalone.py 0 200 [budget_s]        (NMX_FUZZ_EMU=1: the CPU logic emulator)"""
import os
import sys
import time
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
warnings.filterwarnings("ignore")


def one(seed):
    import pandas as pd
    from scipy.signal import fftconvolve

    from oracle import mne_restated as mr
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings, features, fir_design
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.processing import NotchFilter, PreprocessingFilter, ReReferencer, Resampler

    rng = np.random.default_rng(seed)
    big = 60000 if not os.environ.get("NMX_FUZZ_EMU") else 24000
    kind = seed % 5
    C = int(rng.integers(1, 5))
    if kind == 0:
        fs = float(rng.choice([250.0, 500.0, 1000.0, 2000.0]))
        T = int(rng.integers(int(fs) // 2, big))
        x = rng.standard_normal((C, T)) * rng.uniform(1, 100) + rng.uniform(-500, 500, (C, 1))
        nf = NotchFilter(fs, 50)
        arg = x[0] if C == 1 and seed % 2 else x
        got, want = nf.process(arg), orc.NotchFilter(fs, 50, taps=nf.filter_bank).process(arg)
    elif kind == 1:
        fs = float(rng.choice([1000.0, 4000.0]))
        T = int(rng.integers(int(fs), big))
        x = rng.standard_normal((C, T)) * rng.uniform(1, 100) + rng.uniform(-500, 500, (C, 1))
        bands = [[4, 8], [8, 12], [13, 35], [60, 200]][: int(rng.integers(1, 5))]
        f = features.MNEFilter(bands, fs, filter_length="999ms")
        got = f.filter_data(x)
        want = np.stack([np.stack([fftconvolve(r, np.asarray(tp, float), "same") for tp in f.filter_bank]) for r in x])
    elif kind == 2:
        fs = 1000.0
        T = int(rng.integers(1000, big))
        x = rng.standard_normal((C, T)) * rng.uniform(1, 100) + rng.uniform(-500, 500, (C, 1))
        s = NMSettings.get_default()
        pfs = s.preprocessing_filter
        on = rng.random(4) < 0.5
        if not on.any():
            on[int(rng.integers(0, 4))] = True
        pfs.bandstop_filter, pfs.bandpass_filter, pfs.lowpass_filter, pfs.highpass_filter = (bool(v) for v in on)
        s = s.validate()
        got = PreprocessingFilter(s, fs).process(x)
        want = orc.PreprocessingFilter(s, fs, taps=fir_design.preprocessing_filter_bank(s.preprocessing_filter, fs)).process(x)
    elif kind == 3:
        fs = float(rng.choice([1000.0, 2000.0, 4000.0, 1375.0, 22050.0, 512.0, 250.0, 30000.0]))
        to = float(rng.choice([1000.0, 250.0, 500.0, 999.0, 1024.0, 3000.0, 128.0]))
        if fs == to:
            to = fs / 2
        T = int(rng.integers(1, 4 * big))
        x = rng.standard_normal((C, T)) * rng.uniform(0.1, 100) + rng.uniform(-1e3, 1e3)
        got, want = Resampler(fs, to).process(x), mr.resample(x, up=to / fs, down=1.0)
        assert got.shape == want.shape
        if want.size:
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * np.abs(want).max())
        return
    else:
        n = int(rng.integers(2, 80))
        names = [f"c{i}" for i in range(n)]
        types = [str(rng.choice(["ecog", "dbs", "seeg"])) for _ in range(n)]
        ref = []
        for i in range(n):
            r = rng.random()
            others = [m for m in names if m != names[i]]
            ref.append("average" if r < 0.5 else "None" if r < 0.6 else "&".join(rng.choice(others, size=min(len(others), int(rng.integers(1, 3))),
                                                                                          replace=False)))
        ch = pd.DataFrame({"name": names, "rereference": ref, "used": [1] * n, "target": [0] * n, "type": types,
                           "status": ["good" if rng.random() < 0.9 else "bad" for _ in range(n)], "new_name": names})
        try:
            rr = ReReferencer(1000.0, ch)
        except ZeroDivisionError:   # an average over an empty group: the reference divides by len([]) too (rereference.py:79)
            return
        R = chmod.reref_matrix(chmod.load_channels(ch))
        if R is None:
            return
        x = rng.standard_normal((R.shape[1], int(rng.integers(1, big)))) * 50 + rng.uniform(-4000, 4000, (R.shape[1], 1))
        got, want = rr.process(x), R @ x
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-10)
        return
    assert got.shape == want.shape, (got.shape, want.shape)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * np.abs(want).max())


def main(lo, hi, budget_s=600.0):
    from py_neuromodulation_amd import _lib

    if os.environ.get("NMX_FUZZ_EMU"):
        import __graft_entry__ as ge
        _lib._default = _lib.NmxLibrary(ge.build_emu())
    t0, n, bad = time.time(), 0, 0
    for seed in range(lo, hi):
        if time.time() - t0 > budget_s:
            print(f"time budget reached at seed {seed}")
            break
        n += 1
        try:
            one(seed)
        except BaseException as e:   # noqa: BLE001
            bad += 1
            print(f"FAIL kind {seed % 5} seed {seed}: {type(e).__name__}: {str(e)[:500]}", flush=True)
    print(f"{n} cases, {bad} failures, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 600.0)

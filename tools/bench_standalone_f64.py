"""The stand-alone float64 pre-processor objects on the GPU (nmx_reref_f64, nmx_resample_f64) next to the host arithmetic
the reference does for the same call (NumPy float64 matmul; scipy FFT resampling as restated in oracle/mne_restated.py --
the oracle is the checker and the timed CPU baseline here, never the product).  One JSON line per case."""

import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from oracle import mne_restated as mr  # noqa: E402
from py_neuromodulation_amd import channels as chmod  # noqa: E402
from py_neuromodulation_amd.processing import ReReferencer, Resampler  # noqa: E402


def timed(fn, n=20):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3)


def main():
    rng = np.random.default_rng(0)
    for fs, to, C, T in ((4000.0, 1000.0, 5, 40000), (1000.0, 4000.0, 5, 10000), (2000.0, 1000.0, 256, 2000),
                         (1375.0, 1000.0, 64, 13750), (4000.0, 1000.0, 256, 40000)):
        x = rng.standard_normal((C, T)) * 10 + 100
        r = Resampler(fs, to)
        got, want = r.process(x), mr.resample(x, up=to / fs)
        print(json.dumps({"case": f"Resampler {fs:g}->{to:g} Hz, {C} ch x {T} samples", "gpu_ms": round(timed(lambda: r.process(x)), 3),
                          "cpu_scipy_ms": round(timed(lambda: mr.resample(x, up=to / fs), 3), 3),
                          "max_abs_err_rel_to_max": float(np.abs(got - want).max() / np.abs(want).max())}), flush=True)
    import pandas as pd

    for C, T in ((6, 1000), (256, 1000), (256, 40000), (1024, 10000)):
        names = [f"c{i}" for i in range(C)]
        ch = pd.DataFrame({"name": names, "rereference": ["average"] * C, "used": [1] * C, "target": [0] * C,
                           "type": ["ecog"] * (C // 2) + ["dbs"] * (C - C // 2), "status": ["good"] * C, "new_name": names})
        rr = ReReferencer(1000.0, ch)
        R = chmod.reref_matrix(chmod.load_channels(ch))
        x = rng.standard_normal((C, T)) * 50 + rng.uniform(-4000, 4000, (C, 1))
        got, want = rr.process(x), R @ x
        print(json.dumps({"case": f"ReReferencer average, {C} ch x {T} samples", "gpu_ms": round(timed(lambda: rr.process(x)), 3),
                          "cpu_numpy_matmul_ms": round(timed(lambda: R @ x, 5), 3),
                          "max_abs_err_rel_to_max": float(np.abs(got - want).max() / np.abs(want).max())}), flush=True)


if __name__ == "__main__":
    main()

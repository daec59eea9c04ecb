"""cProfile of Stream(...) construction on a warm process (256 ch x 120 s float64, default settings)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import py_neuromodulation_amd as nm  # noqa: E402

C, T = 256, 120000
rng = np.random.default_rng(0)
data = rng.standard_normal((C, T)) * 50 + rng.uniform(-300, 300, (C, 1))
for _ in range(3):
    st = nm.Stream(sfreq=1000, data=data)
    st.run(save_csv=False)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    st = nm.Stream(sfreq=1000, data=data)
    ts.append(time.perf_counter() - t0)
print("construct ms", [round(1e3 * t, 2) for t in ts])
pr = cProfile.Profile()
pr.enable()
st = nm.Stream(sfreq=1000, data=data)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

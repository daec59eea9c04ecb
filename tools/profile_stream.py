#!/usr/bin/env python
"""Where Stream.run's wall time goes (256 ch x 120 s float64 recording, default settings): cProfile of the
third run, top cumulative entries."""
import cProfile
import pstats
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import py_neuromodulation_amd as nm

C, T = 256, 120000
rng = np.random.default_rng(0)
data = rng.standard_normal((C, T)) * 50 + rng.uniform(-300, 300, (C, 1))
for _ in range(2):
    nm.Stream(sfreq=1000, data=data).run(save_csv=False)
pr = cProfile.Profile()
pr.enable()
df = nm.Stream(sfreq=1000, data=data).run(save_csv=False)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)

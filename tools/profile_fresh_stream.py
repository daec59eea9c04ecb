"""cProfile of Stream.run on 256 ch x 120 s float64: a FRESH Stream object on a warm process, then the same object again
(where the host-side milliseconds around the library call go)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import py_neuromodulation_amd as nm
C, T = 256, 120000
rng = np.random.default_rng(0)
data = rng.standard_normal((C, T)) * 50 + rng.uniform(-300, 300, (C, 1))
for _ in range(3):
    st = nm.Stream(sfreq=1000, data=data); st.run(save_csv=False)
st = nm.Stream(sfreq=1000, data=data)
pr = cProfile.Profile(); pr.enable(); t0=time.perf_counter(); st.run(save_csv=False); t1=time.perf_counter(); pr.disable()
print("fresh run", t1-t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).print_callees("run_pipelined")
pstats.Stats(pr).print_callees("process_batch_f64")
pr = cProfile.Profile(); pr.enable(); t0=time.perf_counter(); st.run(save_csv=False); t1=time.perf_counter(); pr.disable()
print("same object again", t1-t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

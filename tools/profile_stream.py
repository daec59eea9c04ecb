#!/usr/bin/env python
"""Where Stream.run's wall time goes (256 ch x 120 s float64 recording, default settings): phase timers around the
pieces of the run, then cProfile of a warm run (top cumulative entries)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import py_neuromodulation_amd as nm
from py_neuromodulation_amd.engine import parallel_cast

C, T = 256, 120000
rng = np.random.default_rng(0)
data = rng.standard_normal((C, T)) * 50 + rng.uniform(-300, 300, (C, 1))
st = nm.Stream(sfreq=1000, data=data)
for _ in range(2):
    st.run(save_csv=False)
dp = st.data_processor
eng = dp.engine
starts = np.arange(0, T - 1000 + 1, 100)


def timed(label, fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    print(f"{label:58s} {1e3 * min(ts):8.2f} ms (best of {n})")
    return r


x32 = eng.pinned_empty(data.shape)
timed("cast float64 -> pinned float32 (parallel_cast)", lambda: parallel_cast(x32, data))
out = eng.pinned_empty((len(starts), eng.n_outputs))
dp.reset()
timed("engine.process_batch, pinned float32 in / pinned out", lambda: eng.process_batch(x32, starts, out=out))
timed("engine.process_batch, float64 in (cast + batch)", lambda: eng.process_batch(data, starts, staged_output=True))
timed("engine.process_batch_f64 (pipelined)", lambda: eng.process_batch_f64(data, starts, want_nan_mask=True))
o64 = np.empty(out.shape)
timed("widen float32 -> float64 table", lambda: parallel_cast(o64, out))
import pandas as pd
keys = list(dp.keys)
df = timed("DataFrame(rows, columns=keys)", lambda: pd.DataFrame(o64, columns=keys))
timed("df['time'] = ...", lambda: df.__setitem__("time", np.arange(len(df), dtype=float)), n=1)
timed("dp.reset()", dp.reset)
timed("_save_after_stream", lambda: st._save_after_stream("", "sub"))
timed("Stream.run(save_csv=False)", lambda: st.run(save_csv=False))
timed("Stream(...) construction, warm", lambda: nm.Stream(sfreq=1000, data=data))
pr = cProfile.Profile()
pr.enable()
st.run(save_csv=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

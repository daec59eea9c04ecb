import sys, time; sys.path.insert(0, ".")
import numpy as np
import py_neuromodulation_amd as nm
from py_neuromodulation_amd import data_processor as dpm, engine as em
C, T = 256, 120000
rng = np.random.default_rng(0)
data = rng.standard_normal((C, T)) * 50 + rng.uniform(-300, 300, (C, 1))
st = nm.Stream(sfreq=1000, data=data); st.run(save_csv=False); st.run(save_csv=False)
marks = []
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); marks.append((name, time.perf_counter() - t0)); return r
    setattr(cls, name, g)
wrap(dpm.DataProcessor, "process_batch"); wrap(em.HotPathEngine, "process_batch_f64"); wrap(em.HotPathEngine, "run_pipelined"); wrap(dpm.DataProcessor, "reset")
for label, fresh in (("same object", False), ("fresh object", True), ("same object", False), ("fresh object", True)):
    marks.clear()
    if fresh:
        t0 = time.perf_counter(); st = nm.Stream(sfreq=1000, data=data); tc = time.perf_counter() - t0
    else:
        tc = 0.0
    t0 = time.perf_counter(); st.run(save_csv=False); tr = time.perf_counter() - t0
    print(label, "construct %.2f ms run %.2f ms" % (tc * 1e3, tr * 1e3), [(n, round(v * 1e3, 2)) for n, v in marks])

#!/bin/bash
# page-locked host-memory batch of the headline workload (1024 hops x 256 ch): first chunk / following chunks
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for fb in "128 512" "256 768" "256 1024" "128 1024" "192 832" "320 704" "512 512" "384 640"; do
  set -- $fb
  NMX_HOST_FIRST_CHUNK=$1 NMX_HOST_CHUNK_WINDOWS=$2 python - <<PY
import sys, time, numpy as np
sys.path.insert(0, ".")
import bench
from py_neuromodulation_amd import fir_design
from py_neuromodulation_amd.engine import HotPathEngine
s = bench.make_settings(); C, W, hop, n = 256, 1000, 100, 1024
T = W + (n - 1) * hop
eng = HotPathEngine(s, [f"ch{i}_avgref" for i in range(C)], 1000.0, ref_matrix=bench.car_matrix(C), notch_taps=fir_design.notch_bank(1000.0, 50))
xp = eng.pinned_empty((C, T)); xp[...] = bench.synth(C, T, 1000.0, 1)
op = eng.pinned_empty((n, eng.n_outputs)); starts = np.arange(n, dtype=np.int64) * hop
for _ in range(3): eng.process_batch(xp, starts, out=op)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); eng.process_batch(xp, starts, out=op); ts.append(time.perf_counter() - t0)
print("first $1 then $2: median %.3f ms, best %.3f ms per 1024 hops" % (1e3 * float(np.median(ts)), 1e3 * min(ts)))
PY
done
done

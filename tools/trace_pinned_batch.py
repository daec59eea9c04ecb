#!/usr/bin/env python
"""A few host-memory batches of the headline workload with page-locked buffers both ways (what tools/bench_host.py times as
`pcie_inclusive_windows_per_s`), for `rocprofv3 --kernel-trace --memory-copy-trace`:  python tools/trace_pinned_batch.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def main():
    from py_neuromodulation_amd import fir_design
    from py_neuromodulation_amd.engine import HotPathEngine

    s = bench.make_settings()
    C, W, hop, n = 256, 1000, 100, 1024
    T = W + (n - 1) * hop
    eng = HotPathEngine(s, [f"ch{i}_avgref" for i in range(C)], 1000.0, ref_matrix=bench.car_matrix(C),
                        notch_taps=fir_design.notch_bank(1000.0, 50))
    xp = eng.pinned_empty((C, T))
    xp[...] = bench.synth(C, T, 1000.0, 1)
    op = eng.pinned_empty((n, eng.n_outputs))
    starts = np.arange(n, dtype=np.int64) * hop
    for _ in range(4):
        eng.process_batch(xp, starts, out=op)
    time.sleep(0.02)   # (a gap the timeline script finds the last batch by)
    t0 = time.perf_counter()
    eng.process_batch(xp, starts, out=op)
    print(f"last batch {1e3 * (time.perf_counter() - t0):.3f} ms")


if __name__ == "__main__":
    main()

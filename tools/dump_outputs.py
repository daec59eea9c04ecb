#!/usr/bin/env python
"""Features of a short bench-shaped batch (64 ch x 80 hops: persistent kernels) to a .npy -- for bit-equality
checks between build variants / knobs (tools/exp_variants.sh)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from py_neuromodulation_amd import fir_design
from py_neuromodulation_amd.engine import HotPathEngine

s = bench.make_settings()
C, n = 64, 80
x = bench.synth(C, 1000 + (n - 1) * 100, 1000.0, 7)
eng = HotPathEngine(s, [f"ch{i}_avgref" for i in range(C)], 1000.0, ref_matrix=bench.car_matrix(C),
                    notch_taps=fir_design.notch_bank(1000.0, 50))
out = eng.process_batch(x, np.arange(n) * 100)
np.save(sys.argv[1], out)
print(out.shape, float(np.nanmean(out)))

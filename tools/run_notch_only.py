#!/usr/bin/env python
"""The notch kernel alone on the bench shape (256 ch, 1024 hops, 999 taps), for counter / per-phase profile runs
(tools/exp_profile_notch.sh)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from py_neuromodulation_amd import NMSettings, fir_design
from py_neuromodulation_amd.engine import HotPathEngine

C, W, n, hop = 256, 1000, 1024, 100
T = W + (n - 1) * hop
dev = torch.device("cuda", 0)
x = torch.randn((C, T), dtype=torch.float32, device=dev) * 50
starts = np.arange(n, dtype=np.int64) * hop
s = NMSettings.get_default()
s.features.disable_all()
s.features.return_raw = True
eng = HotPathEngine(s, [f"ch{i}" for i in range(C)], 1000.0, notch_taps=fir_design.notch_bank(1000.0, 50))
out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
for i in range(3):
    eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize(dev)
print("notch ms (prep stage)", eng.timing_ms(1))

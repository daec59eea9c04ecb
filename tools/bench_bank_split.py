#!/usr/bin/env python
"""How the FIR-bank kernel's time splits: the bench workload (256 ch, 1024 hops of 100 ms, W = 1000, no
pre-processing) with subsets of the features that need filters; HIP-event time of the bank launch.
    python tools/bench_bank_split.py"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import torch

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    C, W, n, hop = 256, 1000, 1024, 100
    T = W + (n - 1) * hop
    dev = torch.device("cuda", 0)
    x = torch.randn((C, T), dtype=torch.float32, device=dev) * 50
    starts = np.arange(n, dtype=np.int64) * hop
    stream = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for name, feats, bands in (("bandpass, 1 band", "bandpass_filter", 1), ("bandpass, 2 bands", "bandpass_filter", 2),
                               ("bandpass, 4 bands", "bandpass_filter", 4), ("bursts (2 filters + series)", "bursts", 4),
                               ("sharpwave (2 filters + series)", "sharpwave_analysis", 4),
                               ("bandpass + bursts (4 filters)", "bandpass_filter,bursts", 4),
                               ("all (6 filters)", "bandpass_filter,bursts,sharpwave_analysis", 4)):
        s = NMSettings.get_default()
        s.features.disable_all()
        for f in feats.split(","):
            setattr(s.features, f, True)
        if bands < 4:
            keep = list(s.frequency_ranges_hz)[2:2 + bands] if bands <= 2 else list(s.frequency_ranges_hz)[:bands]
            s.frequency_ranges_hz = {k: s.frequency_ranges_hz[k] for k in keep}
        eng = HotPathEngine(s, [f"ch{i}" for i in range(C)], 1000.0)
        out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
        ms = []
        for i in range(6):
            eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)
            torch.cuda.synchronize(dev)
            if i >= 2:
                ms.append(eng.timing_ms(3))
        res[name] = round(float(np.mean(ms)), 3)
        del eng
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Where the time-domain / spectral kernel's time goes: the bench workload (256 ch, 1024 hops of 100 ms,
W = 1000) with feature subsets, HIP-event time of the nmx_kern_timeosc launch for each.
    python tools/bench_timeosc_split.py"""
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
    for feats in ("return_raw", "raw_hjorth,linelength,return_raw", "fft", "welch", "stft", "fft,welch",
                  "fft,welch,stft", "raw_hjorth,linelength,return_raw,fft,welch,stft"):
        s = NMSettings.get_default()
        s.features.disable_all()
        for f in feats.split(","):
            setattr(s.features, f, True)
        eng = HotPathEngine(s, [f"ch{i}" for i in range(C)], 1000.0)
        out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
        ms = []
        for i in range(7):
            eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)
            torch.cuda.synchronize(dev)
            if i >= 2:
                ms.append(eng.timing_ms(2))
        res[feats] = float(np.mean(ms))
        del eng
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

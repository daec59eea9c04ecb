#!/usr/bin/env python
"""Roofline of the HBM-bound member of the path (BASELINE config[1] feature set: FFT band power +
Hjorth + LineLength, no pre-processing) in SURVEY 8(d) "Mode A": distinct data per window
(hop = W), input > Infinity Cache, so every byte comes from HBM.
    python tools/bench_scan.py [--channels 256] [--windows 4096] [--features fft,raw_hjorth,linelength]"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--windows", type=int, default=4096)
    ap.add_argument("--features", default="fft,raw_hjorth,linelength")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--burst", type=int, default=4, help="launches back to back per timed sample (the last one is timed)")
    args = ap.parse_args()
    import os

    os.environ.setdefault("NMX_CHUNK_WINDOWS", str(args.windows))   # one launch covers the whole batch
    import torch

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    for f in args.features.split(","):
        setattr(s.features, f, True)
    C, W, n = args.channels, 1000, args.windows
    T = n * W
    eng = HotPathEngine(s, [f"ch{i}" for i in range(C)], 1000.0)
    dev = torch.device("cuda", 0)
    x = torch.randn((C, T), dtype=torch.float32, device=dev) * 50
    out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
    starts = np.arange(n, dtype=np.int64) * W
    stream = torch.cuda.current_stream(dev).cuda_stream
    ms = []
    for i in range(args.steps + 2):
        # --burst launches back to back, the last one timed: its start event is then reached while the GPU still works on
        # the first -- an event recorded on an IDLE stream is stamped before the host has even built the kernel's
        # dispatch packet, and that host latency (~0.1 ms) is not kernel time
        for _ in range(args.burst):
            eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        if i >= 2:
            ms.append(eng.timing_ms(2))   # HIP events around the single nmx_kern_timeosc launch
    t = float(np.mean(ms))
    F_c = eng.n_outputs / C
    nbytes = n * C * (4 * W + 4 * F_c)
    print(json.dumps({"kernel": "nmx_kern_timeosc", "features": args.features, "channels": C, "windows": n,
                      "ms": t, "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / t / 1e6,
                      "frac_of_8TBps": nbytes / t / 1e6 / 8000.0, "input_GB": C * T * 4 / 1e9}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Throughput of the other BASELINE.json configurations on ONE GPU (their per-GPU share), inputs resident
in HBM.  These shapes run on the GENERIC kernels (any window / filter length), not on the one-wave kernels
of the default 1 kHz / 1 s shape:
  C2  64 ch @ 1 kHz, W = 1000, hop 100: FFT band power + Hjorth + LineLength          (config[1])
  C3  256 ch @ 2 kHz, W = 2000, hop 200: 8-band band-pass bank + STFT + bursts         (config[2])
  C4  256 ch @ 1 kHz (one of 4 shards of 1024 ch): oscillatory + sharp waves + notch   (config[3])
  C5  512 ch @ 30 kHz (one of 8 shards of 4096 ch), W = 512, hop 30: full set          (config[4])
    python tools/bench_configs.py [C3 C5 ...]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def run(eng, C, W, hop, n, dev, torch, steps=12):
    T = W + (n - 1) * hop
    x = torch.randn((C, T), dtype=torch.float32, device=dev) * 50
    out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
    starts = np.arange(n, dtype=np.int64) * hop
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):
        eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, st)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, st)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    stages = (("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6), ("bursts", 4), ("sharp", 5))
    return {"windows_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "hops_per_batch": n,
            "channels": C, "features_per_window": eng.n_outputs, "nan_outputs": int(torch.isnan(out).sum().item()),
            "stage_ms": {k: round(eng.timing_ms(i), 3) for k, i in stages},
            "kernels": {k: eng.kernels(i) for k, i in stages if eng.kernels(i)}}


def main():
    import torch

    from py_neuromodulation_amd import NMSettings, fir_design
    from py_neuromodulation_amd.engine import HotPathEngine

    dev = torch.device("cuda", 0)
    res = {}
    only = [a.upper() for a in sys.argv[1:]]   # e.g. "C3 C5": just these configurations

    def want(tag):
        return not only or tag in only

    # C2
    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = True
    if want("C2"):
        eng = HotPathEngine(s, [f"ch{i}" for i in range(64)], 1000.0)
        res["C2 64ch@1kHz fft+hjorth+linelength"] = run(eng, 64, 1000, 100, 4096, dev, torch)
        eng.close()
    # C3
    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = s.features.stft = s.features.bursts = True
    s.frequency_ranges_hz = {"theta": [4, 8], "alpha": [8, 12], "low_beta": [13, 20], "high_beta": [20, 35],
                             "low_gamma": [60, 80], "high_gamma": [90, 200], "HFA": [200, 400], "broadband": [4, 400]}
    s.bandpass_filter_settings.segment_lengths_ms["broadband"] = 1000
    s = s.validate()
    if want("C3"):
        eng = HotPathEngine(s, [f"ch{i}" for i in range(256)], 2000.0)
        res["C3 256ch@2kHz 8-band bank+stft+bursts"] = run(eng, 256, 2000, 200, 256, dev, torch)
        eng.close()
    # C4 (one shard)
    s = NMSettings.get_default()
    s.features.disable_all()
    for f in ("fft", "welch", "stft", "bandpass_filter", "sharpwave_analysis"):
        setattr(s.features, f, True)
    if want("C4"):
        eng = HotPathEngine(s, [f"ch{i}" for i in range(256)], 1000.0, notch_taps=fir_design.notch_bank(1000.0, 50))
        res["C4 shard 256ch@1kHz oscillatory+sharpwave+notch"] = run(eng, 256, 1000, 100, 1024, dev, torch)
        eng.close()
    # C5 (one shard)
    base = NMSettings.get_default().to_dict()
    base["frequency_ranges_hz"] = {"gamma": [60, 200], "HFA": [200, 500], "MUA": [500, 3000], "spike": [3000, 7000]}
    s = NMSettings(**base)
    s.features.disable_all()
    for f in ("fft", "stft", "raw_hjorth", "linelength", "return_raw", "bandpass_filter", "sharpwave_analysis"):
        setattr(s.features, f, True)
    s.sampling_rate_features_hz = 1000
    s.segment_length_features_ms = 17
    s.fft_settings.windowlength_ms = 17
    s.stft_settings.windowlength_ms = 17
    s.bandpass_filter_settings.segment_lengths_ms = {"gamma": 17, "HFA": 10, "MUA": 5, "spike": 3}
    s.sharpwave_analysis_settings.filter_ranges_hz = [[500, 3000], [1000, 7000]]
    s = NMSettings(**s.to_dict())
    if want("C5"):
        eng = HotPathEngine(s, [f"c{i}" for i in range(512)], 30000.0, window=512)
        res["C5 shard 512ch@30kHz W=512 hop=30"] = run(eng, 512, 512, 30, 1024, dev, torch)
        eng.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Time / oscillatory kernel of BASELINE config 5 (512 ch @ 30 kHz, 512-sample windows) per feature subset."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import torch

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    dev = torch.device("cuda", 0)
    C, W, hop, n = 512, 512, 30, 1024
    T = W + (n - 1) * hop
    x = torch.randn((C, T), dtype=torch.float32, device=dev) * 50
    starts = np.arange(n, dtype=np.int64) * hop
    res = {}
    for feats in (["fft"], ["stft"], ["raw_hjorth", "linelength", "return_raw"], ["fft", "stft", "raw_hjorth", "linelength", "return_raw"]):
        base = NMSettings.get_default().to_dict()
        base["frequency_ranges_hz"] = {"gamma": [60, 200], "HFA": [200, 500], "MUA": [500, 3000], "spike": [3000, 7000]}
        s = NMSettings(**base)
        s.features.disable_all()
        for f in feats:
            setattr(s.features, f, True)
        s.sampling_rate_features_hz = 1000
        s.segment_length_features_ms = 17
        s.fft_settings.windowlength_ms = 17
        s.stft_settings.windowlength_ms = 17
        s = NMSettings(**s.to_dict())
        eng = HotPathEngine(s, [f"c{i}" for i in range(C)], 30000.0, window=512)
        out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(3):
            eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, st)
        torch.cuda.synchronize(dev)
        res["+".join(feats)] = {"timeosc_ms": round(eng.timing_ms(2), 3), "kernel": eng.kernels(2), "n_out": eng.n_outputs}
        eng.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""raw_normalization on the device at the headline width: 256 ch @ 1 kHz, 30 s history, 1024 hops per batch.
    python tools/bench_rawnorm.py [method ...]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import torch

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    dev = torch.device("cuda", 0)
    C, W, hop, n = 256, 1000, 100, 1024
    T = W + (n - 1) * hop
    x = torch.randn((C, T), dtype=torch.float32, device=dev) * 50
    starts = np.arange(n, dtype=np.int64) * hop
    res = {}
    for method in (sys.argv[1:] or ["zscore", "median", "zscore-median", "robust", "minmax"]):
        eng = HotPathEngine(NMSettings.get_default(), [f"c{i}" for i in range(C)], 1000.0, features=["return_raw"],
                            raw_norm=(method, 3, 30000, 100), window=W)
        out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        ms = []
        for _ in range(4):   # the first batches fill the 30 s history
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, st)
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t0) * 1e3)
        res[method] = {"ms_per_1024_hops": [round(m, 2) for m in ms], "hops_per_s_steady": round(n / (ms[-1] * 1e-3), 1),
                       "kernels": eng.kernels(1)}
        eng.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

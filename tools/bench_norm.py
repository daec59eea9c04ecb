#!/usr/bin/env python
"""Time the device feature normaliser (nmx_norm_process) on a bench-sized feature matrix."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.processing import DeviceFeatureNormalizer

    n, F = 1024, 9984
    s = NMSettings.get_default()
    if len(sys.argv) > 1:
        s.feature_normalization_settings.normalization_method = sys.argv[1]   # mean | zscore | median | zscore-median
    dn = DeviceFeatureNormalizer(s, F)
    x = torch.randn(n, F, device="cuda") * 3 + 1
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        dn.process_device(x.data_ptr(), F, n, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        dn.process_device(x.data_ptr(), F, n, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"method": s.feature_normalization_settings.normalization_method, "rows": n, "features": F, "n_hist": dn.num_samples_normalize, "ms_per_batch": dt * 1e3,
                      "rows_per_s": n / dt, "GBps_rows_rw": 2 * n * F * 4 / dt / 1e9}))


if __name__ == "__main__":
    main()

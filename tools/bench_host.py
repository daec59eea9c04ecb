#!/usr/bin/env python
"""Host-boundary numbers that are NOT bench.py's `value`: PCIe-inclusive batch rate (host buffers
in / out through nmx_process_batch memspace 0) and the latency of the reference's one-window call
shape (DataProcessor.process -> dict) on the bench workload."""
import json
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
    x = bench.synth(C, T, 1000.0, 1)
    starts = np.arange(n, dtype=np.int64) * hop
    eng.process_batch(x, starts)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        eng.process_batch(x, starts)
    dt_pageable = (time.perf_counter() - t0) / reps
    # the same call with page-locked buffers of the caller's (engine.pinned_empty): what the boundary can do
    xp = eng.pinned_empty(x.shape)
    xp[...] = x
    op = eng.pinned_empty((n, eng.n_outputs))
    eng.process_batch(xp, starts, out=op)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.process_batch(xp, starts, out=op)
    dt = (time.perf_counter() - t0) / reps
    lat = []
    w64 = x[:, :W].astype(np.float64)
    for _ in range(30):
        t1 = time.perf_counter()
        out = eng.process_window(w64)
        d = dict(zip(eng.keys, out.tolist()))
        lat.append(time.perf_counter() - t1)
    # the same batch as the float64 table a stream returns (conversion, copies, kernels and widening pipelined:
    # HotPathEngine.process_batch_f64) -- what the multi-device numbers below compare with
    f64_ms = {}
    for name, xx in (("float32_recording", x), ("float64_recording", x.astype(np.float64))):
        eng.process_batch_f64(xx, starts)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.process_batch_f64(xx, starts)
        f64_ms[name] = (time.perf_counter() - t0) / reps * 1e3
    # single process, several plans (Stream(devices=[...])): what each device is handed -- local input (its rows + hi / lo
    # rows per group sum) against the whole recording
    from py_neuromodulation_amd import _lib
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.sharding import MultiDeviceProcessor

    ndev = _lib.get_library().device_count()
    devices = list(range(min(ndev, 8))) if ndev >= 2 else [0, 0]
    channels = chmod.get_default_channels_from_data(x)
    md = {}
    for local in (True, False):
        dp = MultiDeviceProcessor(1000.0, s, channels, line_noise=50, devices=devices, window=W, local_input=local)
        dp.process_batch(x, starts)
        t0 = time.perf_counter()
        for _ in range(reps):
            dp.process_batch(x, starts)
        md["local_input" if dp.local_input else "replicated_input"] = {
            "devices": devices, "h2d_rows_per_device": dp.h2d_rows, "h2d_MB_per_device": [r * T * 4 / 1e6 for r in dp.h2d_rows],
            "ms_per_1024_hops": (time.perf_counter() - t0) / reps * 1e3}
        dp.close()
    print(json.dumps({"multi_device_stream": md, "one_plan_float64_table_ms_per_1024_hops": f64_ms, "pcie_inclusive_windows_per_s": n / dt, "ms_per_1024_hops": dt * 1e3,
                      "pageable_buffers_windows_per_s": n / dt_pageable, "pageable_ms_per_1024_hops": dt_pageable * 1e3,
                      "h2d_MB": x.nbytes / 1e6, "d2h_MB": n * eng.n_outputs * 4 / 1e6,
                      "one_window_256ch_latency_ms_median": float(np.median(lat)) * 1e3,
                      "one_window_features": len(d)}))


if __name__ == "__main__":
    main()

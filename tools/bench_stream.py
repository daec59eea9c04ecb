#!/usr/bin/env python
"""End-to-end drop-in call: py_neuromodulation_amd.Stream(...).run(numpy recording) -> DataFrame, default
settings (feature normalisation on), 256 ch x 120 s @ 1 kHz (1191 hops), wall clock incl. plan creation,
host copies, the device feature normaliser and DataFrame assembly.
    python tools/bench_stream.py [--channels 256] [--seconds 120] [--sfreq 1000]
--sfreq other than 1000: the recording is brought to 1 kHz by the default raw_resampling (windows of `sfreq` samples:
above 7992 through the polyphase path of nmx_k_resample.h), features designed for the new rate
(resample_features_at_new_rate=True)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--sfreq", type=float, default=1000.0)
    args = ap.parse_args()
    import py_neuromodulation_amd as nm

    sf = args.sfreq
    C, T = args.channels, int(args.seconds * sf)
    rng = np.random.default_rng(0)
    t = np.arange(T) / sf
    data = rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + rng.uniform(-300, 300, (C, 1))
    res = {}
    for rep in range(3):
        t0 = time.perf_counter()
        stream = nm.Stream(sfreq=sf, data=data, resample_features_at_new_rate=sf != 1000.0)
        t1 = time.perf_counter()
        df = stream.run(save_csv=False)
        t2 = time.perf_counter()
        res[f"run{rep}"] = {"construct_s": round(t1 - t0, 3), "run_s": round(t2 - t1, 3), "hops": len(df),
                            "columns": df.shape[1], "hops_per_s": round(len(df) / (t2 - t1), 1)}
    # the same Stream object run again and again (what a caller who keeps it sees): best and median of seven
    runs = []
    for _ in range(7):
        t1 = time.perf_counter()
        df = stream.run(save_csv=False)
        runs.append(time.perf_counter() - t1)
    res["warm_same_stream"] = {"run_s_min": round(min(runs), 4), "run_s_median": round(float(np.median(runs)), 4),
                               "hops_per_s_best": round(len(df) / min(runs), 1),
                               "hops_per_s_median": round(len(df) / float(np.median(runs)), 1)}
    res["sfreq"] = sf
    print(json.dumps(res))


if __name__ == "__main__":
    main()

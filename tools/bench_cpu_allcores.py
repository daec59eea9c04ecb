#!/usr/bin/env python
"""CPU baseline on ALL host cores (context row of the result table, never the bench's `cpu_baseline`):
the float64 oracle (restatement of the reference's DataProcessor.process) over the bench workload --
256 ch @ 1 kHz, all nine features, notch + common-average re-referencing -- with the channels split over
P worker processes.  Re-referencing is a per-sample map over all channels: the parent applies it once to
the stream (as SURVEY 8(e) prescribes for shards), the workers run notch + features on their channels.
    python tools/bench_cpu_allcores.py [--procs N] [--hops 8]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _worker(args):
    x, names, hops = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    import bench
    from oracle import nm_oracle as orc

    s = bench.make_settings()
    s.preprocessing = ["raw_resampling", "notch_filter"]          # re-referenced by the parent
    C = len(names)
    channels = {"name": names, "rereference": ["None"] * C, "used": [1] * C, "target": [0] * C,
                "type": ["ecog"] * C, "status": ["good"] * C, "new_name": names}
    dp = orc.DataProcessor(1000.0, s, channels, line_noise=50)
    dp.process(x[:, :1000])
    t0 = time.perf_counter()
    for k in range(1, hops + 1):
        dp.process(x[:, k * 100:k * 100 + 1000])
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--hops", type=int, default=8)
    ap.add_argument("--channels", type=int, default=256)
    a = ap.parse_args()
    import bench

    P = a.procs or min(a.channels, os.cpu_count() or 1)
    C, T = a.channels, 1000 + (a.hops + 1) * 100
    x = bench.synth(C, T, 1000.0, 99).astype(np.float64)
    x = bench.car_matrix(C) @ x                                    # common-average reference, once per sample
    names = [f"ch{i}" for i in range(C)]
    shards = np.array_split(np.arange(C), P)
    jobs = [(x[ix], [names[i] for i in ix], a.hops) for ix in shards if len(ix)]
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        pool.map(_worker, [(j[0][:, :1100], j[1], 1) for j in jobs])   # start the workers, import, design filters
        t0 = time.perf_counter()
        per = pool.map(_worker, jobs)
        wall = time.perf_counter() - t0
    print(json.dumps({"windows_per_s": a.hops / max(per), "pool_wall_s": wall, "procs": len(jobs), "host_cpus": os.cpu_count(),
                      "hops": a.hops, "channels": C, "slowest_worker_s": max(per),
                      "note": "hops / slowest worker's timed loop (workers run concurrently; plan set-up and the first hop are outside the timed loop); oracle float64 port"}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Source bandwidth of the coordinator's staging pass of a multi-device stream (`Stream(devices=[...])`,
sharding.MultiDeviceProcessor): nmx_host_stage_parts reads a float64 recording ONCE and writes every row to its part's
float32 staging array plus the float64 group sums.  GB/s of SOURCE read per thread count -> how many GPUs one
coordinator can feed (a part consumes 256 ch x 100 samples x 8 B = 0.2 MB of source per hop: ~31 GB/s per device at
150 k hops/s).  Host only: needs no GPU.

    python tools/bench_staging.py [channels per part] [seconds]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main(c_part=256, seconds=60):
    from py_neuromodulation_amd import _lib

    lib = _lib.NmxLibrary()
    T = int(seconds * 1000)
    res = {"channels_per_part": c_part, "samples": T, "host_cpus": __import__("os").cpu_count()}
    rng = np.random.default_rng(0)
    for parts in (1, 2, 4, 8):
        C = c_part * parts
        data = rng.standard_normal((C, T))
        xs = [np.empty((c_part, T), np.float32) for _ in range(parts)]
        dst = np.zeros(C, dtype=np.uint64)
        for p, x in enumerate(xs):
            dst[p * c_part:(p + 1) * c_part] = x.ctypes.data + np.arange(c_part, dtype=np.uint64) * np.uint64(x.strides[0])
        gptr = np.array([0, C], dtype=np.int32)          # one common-average group over the whole array
        grows = np.arange(C, dtype=np.int32)
        sums = np.empty(T)
        sum_ptrs = np.array([sums.ctypes.data], dtype=np.uint64)
        row = {}
        for th in (4, 8, 16, 32, 64):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                lib.check(lib.lib.nmx_host_stage_parts(data.ctypes.data, 1, T, C, 0, T, dst.ctypes.data, 1, gptr.ctypes.data,
                                                       grows.ctypes.data, sum_ptrs.ctypes.data, th))
                best = min(best, time.perf_counter() - t0)
            row[f"{th}_threads_GBps_source"] = round(data.nbytes / best / 1e9, 1)
        np.testing.assert_allclose(sums[:100], data[:, :100].astype(np.float32).astype(np.float64).sum(0), rtol=1e-12)
        res[f"{parts}_parts_{C}ch"] = row
        print(parts, row, file=sys.stderr)
        del data, xs
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256, float(sys.argv[2]) if len(sys.argv) > 2 else 60)

#!/usr/bin/env python
"""Does the headline step depend on whether every step re-reads the SAME 1024 hops (bench.py) or continues the stream
with new data (what Stream.run sees)?  The threshold walk keeps a top-K list of the history: repeated data only meets ties."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import torch

    import bench
    from py_neuromodulation_amd import fir_design
    from py_neuromodulation_amd.engine import HotPathEngine

    dev = torch.device("cuda", 0)
    s = bench.make_settings()
    C, W, hop, n = 256, 1000, 100, 1024
    steps, warm = 10, 4
    K = steps + warm
    T1 = W + (n - 1) * hop
    Tall = W + (n * K - 1) * hop
    ch = [f"ch{i}_avgref" for i in range(C)]
    res = {}
    for mode in ("same", "fresh", "same", "fresh"):
        eng = HotPathEngine(s, ch, 1000.0, device=0, ref_matrix=bench.car_matrix(C), notch_taps=fir_design.notch_bank(1000.0, 50))
        x = torch.from_numpy(bench.synth(C, Tall if mode == "fresh" else T1, 1000.0, 1234)).to(dev)
        out = torch.empty((n, eng.n_outputs), dtype=torch.float32, device=dev)
        starts = np.arange(n, dtype=np.int64) * hop
        st = torch.cuda.current_stream(dev).cuda_stream
        ld = x.shape[1]
        kt = {}
        t0 = None
        for i in range(K):
            if i == warm:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            off = i * n * hop if mode == "fresh" else 0
            eng.process_batch_device(x.data_ptr() + 4 * off, ld, ld - off, starts, out.data_ptr(), None, st)
            if i >= warm:
                for name, idx in (("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6), ("bursts", 4), ("sharp", 5)):
                    kt[name] = kt.get(name, 0.0) + eng.timing_ms(idx) / steps
        torch.cuda.synchronize(dev)
        res.setdefault(mode, []).append({"ms_per_step": round((time.perf_counter() - t0) / steps * 1e3, 3),
                                         **{k: round(v, 3) for k, v in kt.items()}})
        eng.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

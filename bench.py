#!/usr/bin/env python
"""bench.py -- windows/sec of the per-hop hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (config.workload): 256 channels per GPU @ 1 kHz, 1 s windows, 100 ms hop, ALL features
of the hot path (raw_hjorth, return_raw, bandpass_filter, stft, fft, welch, sharpwave_analysis,
bursts, linelength; 4 default bands) after notch (50 Hz) + common-average re-referencing.
One step = one batch of --windows hops over synthetic data that is already resident in HBM;
channels shard across GPUs with no collective on the data path (weak scaling: 256 ch / GPU,
each GPU = one independently referenced electrode array).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     dominant kernel (FIR bank): algorithmic bytes / HIP-event kernel time vs 8 TB/s
  cpu_baseline the float64 NumPy/SciPy oracle ("port" of the reference) on the host cores
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def make_settings():
    from py_neuromodulation_amd import NMSettings

    s = NMSettings.get_default()
    s.features.bandpass_filter = True
    s.features.stft = True
    s.postprocessing.feature_normalization = False   # post-processing is outside the kernels
    s.preprocessing = ["raw_resampling", "notch_filter", "re_referencing"]  # resample 1k->1k: no-op
    return s


def synth(C: int, T: int, sfreq: float, seed: int) -> np.ndarray:
    """SURVEY 8(d) generator: 50 N(0,1) + 10 sin(2 pi 20 t) + 5 sin(2 pi 70 t) + dc_c, fp32."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64) / sfreq
    x = rng.standard_normal((C, T), dtype=np.float32) * 50
    x += (10 * np.sin(2 * np.pi * 20 * t) + 5 * np.sin(2 * np.pi * 70 * t)).astype(np.float32)
    x += rng.uniform(-500, 500, size=(C, 1)).astype(np.float32)
    return x


def car_matrix(C: int) -> np.ndarray:
    R = np.full((C, C), -1.0 / (C - 1))
    np.fill_diagonal(R, 1.0)
    return R


def cpu_baseline(s, C: int, sfreq: float, n_windows: int, seed: int):
    """Time the CPU oracle (float64 restatement of the reference's process()) on a bounded sample."""
    from oracle import nm_oracle as orc

    W = int(s.segment_length_features_ms / 1000 * sfreq)
    hop = int(sfreq / s.sampling_rate_features_hz)
    T = W + (n_windows + 1) * hop
    x = synth(C, T, sfreq, seed).astype(np.float64)
    names = [f"ch{i}" for i in range(C)]
    channels = {"name": names, "rereference": ["average"] * C, "used": [1] * C, "target": [0] * C,
                "type": ["ecog"] * C, "status": ["good"] * C, "new_name": [f"{n}_avgref" for n in names]}
    dp = orc.DataProcessor(sfreq, s, channels, line_noise=50)
    dp.process(x[:, :W])  # warm-up (also fills the burst ring like the first hop does)
    t0 = time.perf_counter()
    for k in range(1, n_windows + 1):
        dp.process(x[:, k * hop:k * hop + W])
    dt = time.perf_counter() - t0
    return n_windows / dt, dt


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--channels", type=int, default=256, help="channels per GPU")
    ap.add_argument("--windows", type=int, default=1024, help="hops per step (batch)")
    ap.add_argument("--cpu-windows", type=int, default=24, help="hops timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-preproc", action="store_true", help="skip notch + re-referencing")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; NMX_BENCH_FORCE_DEVICE lets a 1-GPU box exercise the N > 1 code path
    dev_index = int(os.environ.get("NMX_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(args.backend)
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    from py_neuromodulation_amd import fir_design
    from py_neuromodulation_amd.engine import HotPathEngine

    s = make_settings()
    sfreq, C = 1000.0, args.channels
    W = int(s.segment_length_features_ms / 1000 * sfreq)
    hop = int(sfreq / s.sampling_rate_features_hz)
    n_win = args.windows
    T = W + (n_win - 1) * hop
    ch = [f"ch{i}_avgref" for i in range(C)]
    pre = not args.no_preproc
    eng = HotPathEngine(s, ch, sfreq, device=dev_index,
                        ref_matrix=car_matrix(C) if pre else None,
                        notch_taps=fir_design.notch_bank(sfreq, 50) if pre else None)
    F = eng.n_outputs
    x = torch.from_numpy(synth(C, T, sfreq, 1234 + rank)).to(dev)
    out = torch.empty((n_win, F), dtype=torch.float32, device=dev)
    starts = np.arange(n_win, dtype=np.int64) * hop
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    kt = {k: 0.0 for k in ("prep", "timeosc", "bank", "bursts", "sharp", "batch")}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP-event timers of this launch sequence (recorded on the launch stream inside libnmx)
        for name, idx in (("batch", 0), ("prep", 1), ("timeosc", 2), ("bank", 3), ("bursts", 4), ("sharp", 5)):
            kt[name] += eng.timing_ms(idx)
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
    bad = int(torch.isnan(out).sum().item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = args.steps * n_win * world / dt
        F_c = F / C
        bytes_cw = 4 * W + 4 * F_c                      # fp32 window in + features out
        bank_ms = kt["bank"] / args.steps
        # dominant kernel = FIR bank (nmx_kern_bank_w64_*): reads each (channel, window) once, writes
        # its 4 band-pass features; the filtered-series hand-off to the Hilbert / sharp-wave kernels
        # is NOT algorithmic traffic (it shows up in `traffic`)
        bank_bytes = n_win * C * (4 * W + 4 * 4)
        achieved = bank_bytes / (bank_ms * 1e-3) / 1e9 if bank_ms > 0 else 0.0
        traffic = None
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get("nmx_kern_bank_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "windows/sec (all features), 256 ch @ 1 kHz", "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{C}ch/GPU @1kHz, W={W}, hop={hop}, {n_win} hops/step, all 9 hot-path "
                                   f"features, 4 bands, {'notch50+CAR' if pre else 'no preprocessing'}",
                       "channels_per_gpu": C, "windows_per_step": n_win, "features_per_window": F,
                       "parallelism": f"channel-shard x{world}, no collective"},
            "features_per_sec": value * F,
            "algorithmic_GBps_pipeline": value / world * C * bytes_cw / 1e9,
            "kernel_ms_per_step": {k: v / args.steps for k, v in kt.items()},
            "nan_outputs": bad,
            "roofline": {"bound": "hbm", "kernel": "nmx_kern_bank_w64p_scalar<8, 0, 0, 1>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "note": "FIR bank is LDS/FP32-vector bound (SURVEY 8d); frac is vs the HBM roof"},
        }
        if args.cpu_windows > 0 and world == 1:   # CPU baseline: rank 0 at N = 1 only
            v, secs = cpu_baseline(s, C, sfreq, args.cpu_windows, 99)
            res["cpu_baseline"] = {"value": v, "unit": "windows/s", "cores": 1, "kind": "port",
                                   "sample": f"{args.cpu_windows} hops of the same {C}-channel workload "
                                             f"through oracle.DataProcessor.process ({secs:.1f} s)"}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

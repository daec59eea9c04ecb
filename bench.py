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
  roofline     dominant kernel (FIR bank; its NAME comes from the plan = the variant that actually ran):
               algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM roof, the same kernel against the
               157.3 TFLOP/s FP32 vector roof (`roofline.fp32`, SURVEY 8(d): the FIR bank is FP32 / LDS
               bound), and the PMC-measured HBM traffic of that kernel (stamped with the commit it was
               measured at)
  cpu_baseline the float64 NumPy/SciPy oracle ("port" of the reference) on ONE host core, median hop time;
  cpu_baseline_allcores  the same port with the channels split over worker processes (context row)
  cold_start_ms  the first 1024-hop step of a FRESH plan (the burst history fills during its first 291 hops:
               the fill-regime threshold walk costs more than a whole steady-state step)
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
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same guide: FP32 vector (= FP32 matrix) peak


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
    """Time the CPU oracle (float64 restatement of the reference's process()) on a bounded sample:
    median hop time over `n_windows` hops after 2 warm-up hops, one core."""
    from oracle import nm_oracle as orc

    W = int(s.segment_length_features_ms / 1000 * sfreq)
    hop = int(sfreq / s.sampling_rate_features_hz)
    T = W + (n_windows + 2) * hop
    x = synth(C, T, sfreq, seed).astype(np.float64)
    names = [f"ch{i}" for i in range(C)]
    channels = {"name": names, "rereference": ["average"] * C, "used": [1] * C, "target": [0] * C,
                "type": ["ecog"] * C, "status": ["good"] * C, "new_name": [f"{n}_avgref" for n in names]}
    dp = orc.DataProcessor(sfreq, s, channels, line_noise=50)
    rows = []
    for k in range(2):   # warm-up (the first hop also fills the burst ring like the reference's first hop)
        rows.append(dp.process(x[:, k * hop:k * hop + W]))
    per = []
    for k in range(2, n_windows + 2):
        t0 = time.perf_counter()
        rows.append(dp.process(x[:, k * hop:k * hop + W]))
        per.append(time.perf_counter() - t0)
    return 1.0 / float(np.median(per)), float(np.sum(per)), x, rows


FAMILIES = ("RawHjorth", "_raw", "LineLength", "_fft_", "_welch_", "_stft_", "_bandpass_", "_Sharpwave_", "_bursts_")


def max_rel_err(eng_factory, x64, rows, W, hop):
    """SURVEY 8(d) "max rel err" column: the hops the CPU oracle just processed (cpu_baseline leg, same data, fresh
    state on both sides) through a fresh engine; per feature family |gpu - cpu| / max(|cpu|, m), m = the family's
    median magnitude (log10 features and the re-referenced raw sample pass through zero: for them this is the
    absolute error in units of a typical value): maximum, 99.9th percentile, share of entries above 1e-5, count.
    Informational -- the maxima of the spectral and sharp-wave families are single ill-conditioned entries (a bin
    near a spectral null under log10, an extremum decided by one ulp), which tests/parity.py accepts only on a
    per-entry conditioning report; the gate is tests/."""
    eng = eng_factory()
    n = len(rows)
    got = eng.process_batch(x64.astype(np.float32), np.arange(n, dtype=np.int64) * hop).astype(np.float64)
    keys = list(rows[0].keys())
    assert keys == list(eng.keys), "column order differs from the oracle's"
    want = np.array([[r[k] for k in keys] for r in rows], dtype=np.float64)
    eng.close()
    out = {}
    for fam in FAMILIES:
        sel = np.array([fam in k for k in keys])
        if not sel.any():
            continue
        g, w = got[:, sel], want[:, sel]
        ok = np.isfinite(g) & np.isfinite(w)
        floor = float(np.median(np.abs(w[ok]))) if ok.any() else 0.0
        rel = np.abs(g[ok] - w[ok]) / np.maximum(np.abs(w[ok]), max(floor, 1e-30))
        out[fam.strip("_")] = {"max": float(rel.max()) if rel.size else None,
                               "p999": float(np.quantile(rel, 0.999)) if rel.size else None,
                               "share_above_1e-5": float((rel > 1e-5).mean()) if rel.size else None,
                               "entries": int(rel.size), "nonfinite_mismatch": int((np.isfinite(g) != np.isfinite(w)).sum())}
    return out


# Ceilings of max_rel_err per feature family on the headline workload (48 hops x 256 channels against the float64 oracle):
# (share of entries above 1e-5, maximum), set to what round 6 measured (profiles/r06_bench_1gpu.json) + 25 % / x 2 -- the
# computation is deterministic, the headroom is for a different draw of near-null bins after a code change, not for drift.
# The smooth families sit at 2 - 8e-7; STFT / Welch / sharp waves keep the entries the conditioning reports of
# tests/parity.py explain (bins at a spectral null under log10: 124 of 51 200 STFT entries, 1 Welch entry; extrema decided
# by an ulp: 5 of 76 800).  A run above a ceiling FAILS the bench (exit code 3).
PARITY_CEILINGS = {"RawHjorth": (0.0, 2e-6), "raw": (0.0, 2e-6), "LineLength": (0.0, 2e-6), "fft": (2e-5, 5e-6),
                   "welch": (8e-5, 5e-5), "stft": (3.0e-3, 0.1), "bandpass": (0.0, 2e-6), "Sharpwave": (1e-4, 2e-2),
                   "bursts": (2e-5, 5e-6)}


def parity_gate(err: dict) -> dict:
    bad = {}
    for fam, (lim, lim_max) in PARITY_CEILINGS.items():
        e = err.get(fam)
        if not e:
            continue
        if e["share_above_1e-5"] is not None and e["share_above_1e-5"] > lim:
            bad[fam] = {"share_above_1e-5": e["share_above_1e-5"], "ceiling": lim}
        if e["max"] is not None and e["max"] > lim_max:
            bad.setdefault(fam, {}).update({"max": e["max"], "max_ceiling": lim_max})
        if e["nonfinite_mismatch"]:
            bad.setdefault(fam, {})["nonfinite_mismatch"] = e["nonfinite_mismatch"]
    return {"ok": not bad, "ceilings": {k: {"share_above_1e-5": v[0], "max": v[1]} for k, v in PARITY_CEILINGS.items()},
            "violations": bad}


def _allcores_worker(args):
    x, names, hops = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    from oracle import nm_oracle as orc

    s = make_settings()
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


def cpu_baseline_allcores(C: int, hops: int, procs: int):
    """The same port with the channels split over `procs` worker processes (one core each); the parent
    applies the common-average reference once to the stream.  hops / slowest worker."""
    import multiprocessing as mp

    T = 1000 + (hops + 1) * 100
    x = car_matrix(C) @ synth(C, T, 1000.0, 99).astype(np.float64)
    names = [f"ch{i}" for i in range(C)]
    shards = [ix for ix in np.array_split(np.arange(C), procs) if len(ix)]
    jobs = [(x[ix], [names[i] for i in ix], hops) for ix in shards]
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        pool.map(_allcores_worker, [(j[0][:, :1100], j[1], 1) for j in jobs])   # start-up: imports, filter design
        per = pool.map(_allcores_worker, jobs)
    return hops / max(per), len(jobs), max(per)


def bank_flops_per_item(M: int, n_filters: int, seglens) -> float:
    """forward rFFT(M) + per filter (spectral product 6 (M/2 + 1) + inverse rFFT(M) + tail
    variance 3 seglen); rFFT(M) ~ 2.5 M log2 M."""
    rfft = 2.5 * M * np.log2(M)
    return float(rfft + n_filters * (6 * (M / 2 + 1) + rfft) + 3 * sum(seglens))


def bank_flops_per_item_pair(M: int, n_filters: int, seglens) -> float:
    """The channel-pair kernel (nmx_k_bank_w64c.h): per PAIR of channels one complex FFT(M) forward, per filter a
    real-by-complex spectral product (2 M) and a complex inverse FFT(M), tail variances 3 seglen per channel;
    complex FFT(M) ~ 5 M log2 M.  Returned per item (one channel)."""
    cfft = 5.0 * M * np.log2(M)
    return float(0.5 * (cfft + n_filters * (2 * M + cfft)) + 3 * sum(seglens))


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest round's PMC passes (profiles/r0N_hbm_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command) and the commit they were taken at, or (None, None)."""
    tfiles = sorted((ROOT / "profiles").glob("r[0-9][0-9]_hbm_traffic.json"))
    if not tfiles:
        return None, None
    try:
        doc = json.loads(tfiles[-1].read_text())
        want = kernel.replace("void ", "")
        hit = [v for k, v in doc.get("kernels", {}).items() if k.replace("void ", "") == want]
        if not hit:   # the engine lists "name<NB>", the trace the full instantiation "name<NB, true, false>"
            hit = [v for k, v in doc.get("kernels", {}).items() if "<" in want and k.replace("void ", "").startswith(want.rstrip(">") + ",")]
        if hit:   # only reported for the kernel that ran now
            return hit[0]["hbm_bytes_per_launch"], doc.get("measured_at_commit")
    except Exception:
        pass
    return None, None


def roofline_mode_a(torch, dev, dev_index, channels=256, windows=4096, steps=5):
    """SURVEY 8(d) "Mode A" for the HBM-bound half of the north star (BASELINE config[1]'s feature set: FFT band power +
    Hjorth + LineLength): DISTINCT data per window (hop = W = 1000), 4.2 GB of input -- far beyond the 256 MB
    Infinity Cache -- so every byte comes from HBM; ONE launch of the time / oscillatory kernel over all
    channels x windows items, timed with HIP events on the launch stream (nmx_last_timing_ms stage 2)."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = True
    s.postprocessing.feature_normalization = False
    W, T = 1000, windows * 1000
    old = os.environ.get("NMX_CHUNK_WINDOWS")
    os.environ["NMX_CHUNK_WINDOWS"] = str(windows)   # one launch covers the whole batch (read at plan creation)
    try:
        eng = HotPathEngine(s, [f"ch{i}" for i in range(channels)], 1000.0, device=dev_index)
    finally:
        if old is None:
            del os.environ["NMX_CHUNK_WINDOWS"]
        else:
            os.environ["NMX_CHUNK_WINDOWS"] = old
    x = torch.randn((channels, T), dtype=torch.float32, device=dev) * 50
    out = torch.empty((windows, eng.n_outputs), dtype=torch.float32, device=dev)
    starts = np.arange(windows, dtype=np.int64) * W
    stream = torch.cuda.current_stream(dev).cuda_stream
    ms = []
    for i in range(steps + 2):
        # four launches back to back, the last one timed: an event recorded on an IDLE stream is stamped before the host
        # has built the kernel's dispatch packet (~0.1 ms that is not launch duration), and the first launches behind a
        # host synchronisation run before the clocks have settled (tools/bench_scan.py --burst 2 / 4 / 8 / 16 on one
        # lease: 1.038 / 0.977 / 0.998 / 0.993 ms) -- the headline loop, too, is launches back to back.  The rocprofv3
        # trace of the same command (profiles/r05_kernel_stats.csv) holds every launch, the first of each group included.
        for _ in range(4):
            eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        if i >= 2:
            ms.append(eng.timing_ms(2))
    t = float(np.mean(ms))
    nbytes = windows * channels * (4 * W + 4 * eng.n_outputs / channels)
    kern = eng.kernels(2)
    eng.close()
    del x, out
    traffic, traffic_at = measured_traffic(kern)
    return {"bound": "hbm", "kernel": kern, "achieved": nbytes / t / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / t / 1e6 / HBM_PEAK_GBS, "traffic": traffic, "traffic_measured_at_commit": traffic_at,
            "ms_per_launch": t,
            "algorithmic_bytes_per_launch": nbytes,
            "workload": f"Mode A: {channels} ch x {windows} DISTINCT 1000-sample windows (hop = W), "
                        "FFT band power + Hjorth + LineLength, no pre-processing"}


def config_settings(name: str):
    """BASELINE.json configs[3] / configs[4]."""
    from py_neuromodulation_amd import NMSettings

    if name == "headline":   # --scaling strong: the headline's channels as ONE jointly re-referenced array over the ranks
        s = make_settings()
        s.preprocessing = ["notch_filter", "re_referencing"]
        return s, dict(C_all=None, sfreq=1000.0, W=1000, hop=100, window=None)
    if name == "c4":   # 1024 ch @ 1 kHz, full oscillatory + sharp waves + notch, common average over ALL 1024 rows
        s = NMSettings.get_default()
        s.features.disable_all()
        for f in ("fft", "welch", "stft", "bandpass_filter", "sharpwave_analysis"):
            setattr(s.features, f, True)
        s.preprocessing = ["notch_filter", "re_referencing"]
        s.postprocessing.feature_normalization = False
        return s, dict(C_all=1024, sfreq=1000.0, W=1000, hop=100, window=None)
    base = NMSettings.get_default().to_dict()   # c5: 4096 ch @ 30 kHz, 512-sample windows at a 1 kHz feature rate
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
    s.preprocessing = []
    s.postprocessing.feature_normalization = False
    return NMSettings(**s.to_dict()), dict(C_all=4096, sfreq=30000.0, W=512, hop=30, window=512)


def run_config(args, torch, dist, world, rank, dev, dev_index) -> None:
    """--config c4 | c5: ONE array of C_all channels sharded over the N ranks (strong scaling): each rank holds
    only its own channel block in HBM.  c4 is re-referenced JOINTLY: every step each rank sums its rows per
    sample, ONE all-reduce (RCCL) of that [T] float64 row, then its structured re-reference (own row tap +
    coefficient x the sum row) -- the only exchange step of the path.  c5 has no exchange."""
    from py_neuromodulation_amd.data_processor import DataProcessor
    from py_neuromodulation_amd.sharding import channel_shard

    s, cfg = config_settings(args.config)
    C_all, sfreq, W, hop = cfg["C_all"] or args.channels, cfg["sfreq"], cfg["W"], cfg["hop"]
    n_win = args.windows
    T = W + (n_win - 1) * hop
    names = [f"ch{i}" for i in range(C_all)]
    car = args.config in ("c4", "headline")
    channels = {"name": names, "rereference": ["average" if car else "None"] * C_all, "used": [1] * C_all,
                "target": [0] * C_all, "type": ["ecog"] * C_all, "status": ["good"] * C_all,
                "new_name": [f"{n}_avgref" if car else n for n in names]}
    shard = channel_shard(C_all, world, rank)
    dp = DataProcessor(sfreq, s, channels, line_noise=50, verbose=False, device=dev_index, window=W,
                       channel_subset=shard, local_inputs=True)
    eng = dp.engine
    C = len(shard)
    F = eng.n_outputs
    n_rows = eng.C_in   # own rows (+ the hi / lo rows of the group sum for c4)
    x = torch.empty((n_rows, T), dtype=torch.float32, device=dev)
    x[:C] = torch.from_numpy(synth(C, T, sfreq, 1234 + rank)).to(dev)
    out = torch.empty((n_win, F), dtype=torch.float32, device=dev)
    starts = np.arange(n_win, dtype=np.int64) * hop
    stream = torch.cuda.current_stream(dev).cuda_stream
    sum_dev = torch.device("cpu") if args.backend == "gloo" else dev

    def exchange():   # partial column sums -> all-reduce -> hi / lo sum rows of this rank's input
        part = x[:C].sum(dim=0, dtype=torch.float64)
        if world > 1:
            part = part.to(sum_dev)
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
        part = part.to(device=dev)
        hi = part.to(torch.float32)   # the float64 sum travels as hi + lo float32 rows (channels.split_hi_lo)
        x[C] = hi
        x[C + 1] = (part - hi.to(torch.float64)).to(torch.float32)

    def step():
        if car:
            exchange()
        eng.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    names_t = ("prep", "timeosc", "bank", "bank_sw", "bursts", "sharp", "batch")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt_own = dt
    kt = {k: 0.0 for k in names_t}   # per-stage HIP-event times from a few more steps outside the timed region
    n_kt = min(10, args.steps)
    for _ in range(n_kt):
        step()
        for name, idx in (("batch", 0), ("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6), ("bursts", 4), ("sharp", 5)):
            kt[name] += eng.timing_ms(idx)
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
        every = [torch.zeros(1, dtype=torch.float64, device=tdt.device) for _ in range(world)]
        dist.all_gather(every, torch.tensor([dt_own / args.steps * 1e3], dtype=torch.float64, device=tdt.device))
        rank_ms = [float(t.item()) for t in every]
    else:
        rank_ms = None
    bad = int(torch.isnan(out).sum().item())
    exchange_ms = None
    if car:   # the exchange step alone (outside the timed region): partial sums + all-reduce + hi / lo rows, per step
        torch.cuda.synchronize(dev)
        te = time.perf_counter()
        for _ in range(args.steps):
            exchange()
        torch.cuda.synchronize(dev)
        exchange_ms = (time.perf_counter() - te) / args.steps * 1e3
    if rank == 0:
        value = args.steps * n_win / dt               # windows of the WHOLE array (all ranks work on the same windows)
        stage = max(("timeosc", "bank", "sharp", "prep"), key=lambda k: kt[k])
        idx = {"prep": 1, "timeosc": 2, "bank": 3, "sharp": 5}[stage]
        ms = kt[stage] / n_kt
        algo = n_win * C * (4 * W + 4 * F / C)        # window in + features out, this rank's channels
        res = {
            "metric": ("windows/sec (all features), 256 ch @ 1 kHz" if args.config == "headline"
                       else f"windows/sec, BASELINE config {args.config}"), "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {C_all} ch @ {sfreq:g} Hz as ONE array, W={W}, hop={hop}, {n_win} hops/step, "
                                   f"features {list(eng.enabled)}, preprocessing {list(s.preprocessing)}",
                       "channels_total": C_all, "channels_per_gpu": C, "windows_per_step": n_win,
                       "features_per_window_per_gpu": F,
                       "parallelism": f"channel-shard x{world}" + (", group sum all-reduced per step" if car else ", no collective")},
            "features_per_sec": value * F * world,
            "kernel_ms_per_step": {k: v / n_kt for k, v in kt.items()},
            "ms_per_step_by_rank": rank_ms,   # (N > 1: every rank's own wall time per step; `value` uses the slowest)
            "exchange_ms_per_step": exchange_ms,   # c4: column sums + all-reduce + hi / lo rows (inside the timed step; timed alone here)
            "kernels": {name: eng.kernels(i) for name, i in (("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6), ("sharp", 5))},
            "nan_outputs": bad,
            "roofline": {"bound": "hbm", "kernel": eng.kernels(idx), "stage": stage,
                         "achieved": algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0, "traffic": None,
                         "note": "stage with the largest HIP-event time on rank 0; algorithmic bytes = the step's "
                                 "window-in + features-out bytes of this rank's channels"},
        }
        print(json.dumps(res))
    dp.engine.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="headline", choices=("headline", "c4", "c5"),
                    help="headline = BASELINE metric (default); c4 / c5 = the multi-GPU configs as ONE sharded array")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (the default keeps the timed region above one second)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1 on the headline workload: weak = --channels per GPU (no collective), strong = --channels in "
                         "TOTAL split over the ranks, jointly re-referenced (one all-reduce of the column sums per step)")
    ap.add_argument("--no-gate", action="store_true", help="report max_rel_err_vs_cpu without failing on its ceilings")
    ap.add_argument("--channels", type=int, default=256, help="channels per GPU")
    ap.add_argument("--windows", type=int, default=1024, help="hops per step (batch)")
    ap.add_argument("--cpu-windows", type=int, default=48, help="hops timed for cpu_baseline (0 = skip; ~0.5 s per hop on one core)")
    ap.add_argument("--cpu-procs", type=int, default=32, help="worker processes of cpu_baseline_allcores (0 = skip)")
    ap.add_argument("--no-cold-start", action="store_true", help="skip the cold_start_ms measurement")
    ap.add_argument("--no-normalisation", action="store_true",
                    help="time the headline WITHOUT the reference's default z-score (kernel traces: the normalised plan works in "
                         "384-hop chunks, which would mix with the 1024-hop launches the stage timers and rooflines are quoted on)")
    ap.add_argument("--no-mode-a", action="store_true", help="skip the roofline_modeA measurement (time / oscillatory kernel from HBM)")
    ap.add_argument("--no-preproc", action="store_true", help="skip notch + re-referencing")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()

    import torch

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    forced = "NMX_BENCH_FORCE_DEVICE" in os.environ   # (tests: several ranks on one GPU, gloo only)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU over
        # RCCL) -- never report a 1-GPU number under an N-GPU label
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and not forced:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} HIP device(s) visible on this node")
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if world > 1 and not forced and torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks but only {torch.cuda.device_count()} HIP device(s) visible")
    # one process per GPU; NMX_BENCH_FORCE_DEVICE lets a 1-GPU box exercise the N > 1 code path
    dev_index = int(os.environ.get("NMX_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(args.backend)
    if args.config != "headline" or args.scaling == "strong":
        run_config(args, torch, dist if world > 1 else None, world, rank, dev, dev_index)
        if world > 1:
            dist.destroy_process_group()
        return

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

    cold_ms = None
    if not args.no_cold_start and rank == 0:   # first step of a fresh plan (outside the timed region)
        # ... on a process that has launched a kernel before (code object uploaded, device awake).  What stays in the number:
        # the plan's ~3 GB of hand-off buffers allocated for the first time -- 20 ms on a device this process family has used
        # before, up to 260 ms on the first process after other tenants released the memory (driver side, measured both)
        tiny = HotPathEngine(s, ch[:8], sfreq, device=dev_index, ref_matrix=car_matrix(8) if pre else None,
                             notch_taps=fir_design.notch_bank(sfreq, 50) if pre else None)
        tiny.process_window(np.zeros((8, W)) + np.arange(W)[None, :] % 7)
        tiny.close()
        cold = HotPathEngine(s, ch, sfreq, device=dev_index, ref_matrix=car_matrix(C) if pre else None,
                             notch_taps=fir_design.notch_bank(sfreq, 50) if pre else None)
        torch.cuda.synchronize(dev)
        tc = time.perf_counter()
        cold.process_batch_device(x.data_ptr(), T, T, starts, out.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        cold_ms = (time.perf_counter() - tc) * 1e3
        cold.close()
    # The reference's DEFAULT pipeline ends in the feature normaliser (z-score over the last 30 s of feature rows,
    # default_settings.yaml:69-78, processing/normalization.py:93-111): it runs INSIDE the plan's launch sequence and the
    # headline is timed with it.  (`value_without_normalisation`: the same K steps with the normaliser detached.)
    with_norm = not args.no_normalisation
    dn = None
    if with_norm:
        from py_neuromodulation_amd.processing import DeviceFeatureNormalizer

        sn = make_settings()
        sn.postprocessing.feature_normalization = True
        dn = DeviceFeatureNormalizer(sn, eng.n_outputs, device=dev_index)

    def timed(k):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        barrier()
        # EXACTLY k steps, launched back to back (nothing on the host waits inside the timed region)
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize(dev)
        barrier()
        return time.perf_counter() - t0

    plain_dt = timed(args.steps)   # (also fills the 30 s burst history: steady state for the headline leg)
    if with_norm:
        eng.attach_normalizer(dn)
        dt = timed(args.steps)
        eng.attach_normalizer(None)
    else:
        dt = plain_dt
    # per-stage HIP-event times (recorded on the launch stream inside libnmx): a few more steps OUTSIDE the timed region --
    # reading an event waits for it, which would serialise the host against every step above.  Normaliser detached: a
    # step is then ONE launch sequence of 1024 hops and a stage's timer brackets the kernel(s) of the whole step (with it
    # the plan works in chunks of 384 hops and the timers hold the last chunk's)
    kt = {k: 0.0 for k in ("prep", "timeosc", "bank", "bank_sw", "bursts", "sharp", "batch")}
    n_kt = min(10, args.steps)
    for _ in range(n_kt):
        step()
        for name, idx in (("batch", 0), ("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6), ("bursts", 4), ("sharp", 5)):
            kt[name] += eng.timing_ms(idx)
    dt_own = dt
    rank_ms = None
    if world > 1:
        cdev = dev if args.backend == "nccl" else "cpu"
        tdt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
        # every rank's own wall time per step (the value above uses the slowest): lets a SCALE line be read
        every = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([dt_own / args.steps * 1e3], dtype=torch.float64, device=cdev))
        rank_ms = [float(t.item()) for t in every]
    bad = int(torch.isnan(out).sum().item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = args.steps * n_win * world / dt
        F_c = F / C
        bytes_cw = 4 * W + 4 * F_c                      # fp32 window in + features out
        bank_ms = kt["bank"] / n_kt
        # dominant kernel = FIR bank (nmx_kern_bank_w64_*): reads each (channel, window) once, writes
        # its 4 band-pass features; the filtered-series hand-off to the Hilbert / sharp-wave kernels
        # is NOT algorithmic traffic (it shows up in `traffic`)
        n_bp = sum(1 for i in range(int(eng.desc.n_filters)) if eng.desc.filters[i].bp_seglen > 0)
        bank_bytes = n_win * C * (4 * W + 4 * n_bp)
        achieved = bank_bytes / (bank_ms * 1e-3) / 1e9 if bank_ms > 0 else 0.0
        kernel = eng.kernels(3)   # what the plan launched in the FIR-bank stage of the last step
        traffic, traffic_at = measured_traffic(kernel)
        nf = int(eng.desc.n_filters)
        if "w64c" in kernel:   # the M = 1536 channel-pair kernel took the filters with W + (L - 1) / 2 <= 1536
            sel = [i for i in range(nf) if W + (int(eng.desc.filters[i].n_taps) - 1) // 2 <= 1536]
            nf = len(sel)
            flops_item = bank_flops_per_item_pair(1536, nf, [eng.desc.filters[i].bp_seglen for i in sel])
        else:
            flops_item = bank_flops_per_item(2048, nf, [eng.desc.filters[i].bp_seglen for i in range(nf)])
        tflops = n_win * C * flops_item / (bank_ms * 1e-3) / 1e12 if bank_ms > 0 else 0.0
        res = {
            "metric": "windows/sec (all features), 256 ch @ 1 kHz", "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{C}ch/GPU @1kHz, W={W}, hop={hop}, {n_win} hops/step, all 9 hot-path "
                                   f"features, 4 bands, {'notch50+CAR' if pre else 'no preprocessing'}"
                                   f"{', z-score normalisation' if with_norm else ''}",
                       "channels_per_gpu": C, "windows_per_step": n_win, "features_per_window": F,
                       "parallelism": f"channel-shard x{world}, no collective"},
            "features_per_sec": value * F,
            "algorithmic_GBps_pipeline": value / world * C * bytes_cw / 1e9,
            # (this rank's clock; `value` is the slowest rank's)
            "value_without_normalisation": args.steps * n_win * world / plain_dt,
            "ms_per_step_without_normalisation": plain_dt / args.steps * 1e3,
            "normalisation": ("zscore over 30 s of feature rows inside the plan (default_settings.yaml:69-78)"
                              if with_norm else "none (--no-normalisation)"),
            "kernel_ms_per_step": {k: v / n_kt for k, v in kt.items()},
            "ms_per_step_by_rank": rank_ms,   # (N > 1: every rank's own wall time per step; `value` uses the slowest)
            "nan_outputs": bad,
            "kernels": {name: eng.kernels(idx) for name, idx in
                        (("prep", 1), ("timeosc", 2), ("bank", 3), ("bank_sw", 6), ("bursts", 4), ("sharp", 5))},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_measured_at_commit": traffic_at,
                         "algorithmic_bytes_per_launch": bank_bytes,
                         "fp32": {"flops_per_item": flops_item, "filters": nf, "achieved_TFLOPs": tflops,
                                  "peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS, "frac": tflops / FP32_VECTOR_PEAK_TFLOPS},
                         "note": "the FIR bank is FP32-vector / LDS bound (SURVEY 8d): frac is vs the HBM roof as "
                                 "the contract asks, fp32.frac vs the roof that binds; kernel = the band-pass bank "
                                 "(window in, band-pass features out); filters whose taps are too long for it (the "
                                 "two 1651-tap sharp-wave filters of the default settings) run in a second launch, "
                                 "kernels.bank_sw / kernel_ms_per_step.bank_sw"},
            "cold_start_ms": cold_ms,
            "regime": "steady state: warm-up steps fill the 30 s burst history; cold_start_ms = first step of a fresh plan on a process that has launched a kernel before",
        }
        if world == 1 and not args.no_mode_a:
            try:
                eng.close()
                del x, out
                res["roofline_modeA"] = roofline_mode_a(torch, dev, dev_index)
            except Exception as e:   # never let the context rows break the bench line
                res["roofline_modeA"] = {"error": repr(e)}
        if args.cpu_windows > 0 and world == 1:   # CPU baseline: rank 0 at N = 1 only
            v, secs, x64, orows = cpu_baseline(s, C, sfreq, args.cpu_windows, 99)
            res["cpu_baseline"] = {"value": v, "unit": "windows/s", "cores": 1, "kind": "port",
                                   "sample": f"median hop time over {args.cpu_windows} hops (after 2 warm-up hops) of the "
                                             f"same {C}-channel workload through oracle.DataProcessor.process ({secs:.1f} s)"}
            try:
                res["max_rel_err_vs_cpu"] = max_rel_err(
                    lambda: HotPathEngine(s, ch, sfreq, device=dev_index, ref_matrix=car_matrix(C) if pre else None,
                                          notch_taps=fir_design.notch_bank(sfreq, 50) if pre else None),
                    x64, orows, W, hop)
            except Exception as e:   # never let the context rows break the bench line
                res["max_rel_err_vs_cpu"] = {"error": repr(e)}
            if "error" not in res["max_rel_err_vs_cpu"] and pre and C == 256:
                res["parity_gate"] = parity_gate(res["max_rel_err_vs_cpu"])
            if args.cpu_procs > 0 and C == 256:
                try:
                    procs = min(args.cpu_procs, os.cpu_count() or 1)
                    va, np_, slow = cpu_baseline_allcores(C, 8, procs)
                    res["cpu_baseline_allcores"] = {
                        "value": va, "unit": "windows/s", "cores": np_, "kind": f"port, {np_} processes", "host_cpus": os.cpu_count(),
                        "sample": f"8 hops, channels split over {np_} worker processes (1 core each), common-average "
                                  f"reference applied once by the parent; hops / slowest worker ({slow:.1f} s)"}
                except Exception as e:   # never let the context row break the bench line
                    res["cpu_baseline_allcores"] = {"error": repr(e)}
        print(json.dumps(res))
        if not args.no_gate and not res.get("parity_gate", {"ok": True})["ok"]:
            print("bench.py: max_rel_err_vs_cpu above its ceilings: " + json.dumps(res["parity_gate"]["violations"]), file=sys.stderr)
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Channel sharding of the hot path over the GPUs of one node.

Every feature on the path is per channel; the only cross-channel operator, re-referencing, is a fixed
linear map of the INPUT rows.  Two ways to feed a rank:

* ``local_input=False`` (replicated input): every rank is handed the whole recording and applies rows
  [lo_r, hi_r) of the folded (re-reference x channel-pick) matrix on its GPU (structured kernel: one group
  sum per sample).  No collective on the data path, but N x the host-to-device volume.
* ``local_input=True``: a rank is handed ONLY the input rows of its own channel block (plus the few rows its
  bipolar references name).  The group averages of "average" references (processing/rereference.py:61-63) need
  one number per sample and type group from everybody: each rank sums the members it owns, ONE all-reduce
  (RCCL over xGMI with the "nccl" backend; gloo in the CPU tests) of [n_groups, T] float64 -- 8 bytes per
  sample and group, latency bound -- and every rank continues with its own rows + the sum rows.  This is the
  only exchange step of the path; the NaN mask (one byte per window and channel) is all-gathered so that the
  reference's substring NaN policy sees every channel.

The final gather of the small feature table to rank 0 is control plane (``gather_object``).
"""

from __future__ import annotations

import os

import numpy as np

from . import channels as chmod
from .data_processor import DataProcessor, UserColumns, _LazyNanCols
from .engine import table_empty
from .generator import window_schedule
from .settings import NMSettings


def _force_collectives() -> bool:
    """NMX_FORCE_COLLECTIVES=1: run the exchange step through the process group even when it has ONE rank (a 1-GPU
    box can then exercise the RCCL code path -- device tensors, object collectives -- that a multi-GPU node takes)."""
    import os

    return os.environ.get("NMX_FORCE_COLLECTIVES", "0") == "1"


def channel_shard(n_channels: int, world_size: int, rank: int) -> range:
    """Contiguous block of channels for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_channels, world_size)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def global_keys(sfreq, settings, channels) -> list[str]:
    """Reference column order for ALL channels (no GPU needed)."""
    dp = DataProcessor(sfreq, settings, channels, line_noise=50, verbose=False, dry_run=True)
    return list(dp.keys)


class ShardedStream:
    """One rank of a channel-sharded offline stream."""

    def __init__(self, sfreq, channels, settings=None, line_noise=50, rank: int = 0,
                 world_size: int = 1, device: int | None = None, lib=None, local_input: bool = False) -> None:
        self.sfreq = sfreq
        self.settings = NMSettings.load(settings)
        self.channels = chmod.load_channels(channels)
        self.line_noise = line_noise
        self.rank, self.world_size = rank, world_size
        self.device = rank if device is None else device
        self._lib = lib
        names, self.feature_idx, _ = chmod.channel_info(self.channels)
        self.shard = channel_shard(len(names), world_size, rank)
        self.local_input = local_input
        self._dp = None
        if local_input:   # plan the local rows up front (no GPU needed): callers slice the recording with them
            self._dp = self._processor(None, dry_run=True)
            self.local_rows = list(self._dp.local_rows)
            # an input row is OWNED by the rank whose channel block holds its channel
            self.owned_rows = [self.feature_idx[i] for i in self.shard]

    def _processor(self, window, dry_run=False):
        from . import user_features as _registered

        if _registered:   # a plugin sees the window over ALL channels: only a single-process stream can call it
            raise NotImplementedError(
                f"user-registered features {list(_registered)} need every channel in one process: use "
                "Stream(devices=[...]) (sharding.MultiDeviceProcessor) instead of one rank per GPU")
        return DataProcessor(self.sfreq, self.settings, self.channels, line_noise=self.line_noise, verbose=False,
                             device=self.device, window=window, lib=self._lib, channel_subset=self.shard,
                             local_inputs=self.local_input, dry_run=dry_run)

    def _comm_device(self, group=None):
        """Where the tensors of a collective must live: this rank's GPU under the "nccl" backend (= RCCL, which
        only moves device memory), the host under gloo."""
        import torch
        import torch.distributed as dist

        if str(dist.get_backend(group)).lower() == "nccl":
            torch.cuda.set_device(self.device)   # the object collectives (NaN mask, feature table) stage here too
            return torch.device("cuda", self.device)
        return torch.device("cpu")

    def group_sums(self, local_data: np.ndarray, group=None) -> np.ndarray:
        """[n_groups, T] float64: sum over each group's member rows of nan_to_num(x), partial sums of the rows
        this rank OWNS all-reduced over the ranks (the one exchange step of the sharded path)."""
        import torch
        import torch.distributed as dist

        dp = self._dp
        pos = {j: i for i, j in enumerate(self.local_rows)}
        own = set(self.owned_rows)
        part = np.zeros((len(dp.local_groups), local_data.shape[1]))
        local_data = np.asarray(local_data)
        from . import _lib as _libmod

        lib = self._lib if self._lib is not None else _libmod.get_library()   # (host passes only: no device needed)
        native = (local_data.ndim == 2 and local_data.dtype in (np.float32, np.float64) and local_data.strides[1] == local_data.itemsize
                  and local_data.strides[0] > 0 and local_data.strides[0] % local_data.itemsize == 0)
        for g, members in enumerate(dp.local_groups):
            idx = [pos[int(j)] for j in members if int(j) in own]
            if not idx:
                continue
            if native:   # one pass of libnmx's staging helper: nan_to_num of the float32-rounded samples, float64 sum in row order
                rows = np.ascontiguousarray(idx, dtype=np.int32)
                lib.check(lib.lib.nmx_host_group_sums(part[g].ctypes.data, local_data.ctypes.data, int(local_data.dtype == np.float64),
                                                      local_data.strides[0] // local_data.itemsize, rows.ctypes.data, len(rows), 0,
                                                      local_data.shape[1], 0))
            else:
                part[g] = np.nan_to_num(np.asarray(local_data[idx], np.float32)).astype(np.float64).sum(axis=0)   # the float32 samples the devices see
        if dist.is_available() and dist.is_initialized() and part.size and (
                dist.get_world_size(group) > 1 or _force_collectives()):
            t = torch.from_numpy(part).to(self._comm_device(group))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            part = t.cpu().numpy()
        return part

    def _gather_mask(self, mask_local: np.ndarray, n_all: int, group=None) -> np.ndarray:
        """NaN mask over ALL input rows [n_windows, n_all] from every rank's owned rows."""
        import torch
        import torch.distributed as dist

        pos = {j: i for i, j in enumerate(self.local_rows)}
        mine = (np.asarray(self.owned_rows, np.int64), mask_local[:, [pos[j] for j in self.owned_rows]].astype(np.uint8))
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _force_collectives()):
            parts = [None] * dist.get_world_size(group)
            self._comm_device(group)
            dist.all_gather_object(parts, mine, group=group)
        else:
            parts = [mine]
        full = np.zeros((mask_local.shape[0], n_all), dtype=bool)
        for rows, m in parts:
            full[:, rows] = m.astype(bool)
        return full

    def _run_ragged(self, data, starts, lens, times, group):
        """Ragged window lengths (a non-integer number of samples per segment): one processor per length, the hops in
        order as runs of one length, this rank's state (burst histories, Kalman filters, raw-normaliser histories)
        handed from processor to processor where the length changes (DataProcessor.ragged_*, as Stream.run does)."""
        procs = {int(w): self._processor(int(w)) for w in sorted(set(lens.tolist()))}
        for p in procs.values():
            p.ragged_prepare()
        dp0 = procs[min(procs)]
        x = None
        if self.local_input:
            if data.shape[0] != len(self.local_rows):
                raise ValueError(f"local_input: expected the {len(self.local_rows)} rows ShardedStream.local_rows, "
                                 f"got {data.shape[0]}")
            sums = self.group_sums(data, group)
            x = np.concatenate([np.asarray(data, np.float32)] + [chmod.split_hi_lo(v) for v in sums], axis=0)
        cuts = [0] + [i for i in range(1, len(lens)) if lens[i] != lens[i - 1]] + [len(lens)]
        state, runs = None, []
        for a, b in zip(cuts[:-1], cuts[1:]):
            p = procs[int(lens[a])]
            if state is not None:
                p.ragged_set_state(state)
            if x is None:
                runs.append(p.ragged_run(data, starts[a:b]))
            else:
                o, m = p.engine.process_batch(x, starts[a:b], want_nan_mask=True)
                runs.append((o, m, None))
            state = p.ragged_state()
        if x is None:
            return list(dp0.keys), dp0.ragged_finish(runs), times
        out = np.concatenate([r[0] for r in runs])
        mask_all = self._gather_mask(np.concatenate([r[1] for r in runs]), len(self.channels), group)
        if mask_all.any() and mask_all.shape[1] != len(dp0.ch_names_used):
            raise IndexError("boolean index did not match: NaN handling needs every channel used")
        rows = dp0.postprocess_batch(out, mask_all if mask_all.any() else np.zeros((len(out), len(dp0.ch_names_used)), bool),
                                     normalised=False)
        return list(dp0.keys), rows, times

    def run(self, data: np.ndarray, group=None):
        """-> (local_keys, float64[n_windows, n_local], time_ms) for this rank's channels.
        ``data``: the whole recording [C_all, T], or -- with ``local_input`` -- only its rows
        ``self.local_rows`` (in that order)."""
        st = self.settings
        starts, lens, times = window_schedule(data.shape[1], self.sfreq, st.sampling_rate_features_hz,
                                              st.segment_length_features_ms)
        if len(set(lens.tolist())) > 1:
            return self._run_ragged(data, starts, lens, times, group)
        dp = self._processor(int(lens[0]) if len(lens) else None)
        if not self.local_input:
            rows = dp.process_batch(data, starts) if len(starts) else np.empty((0, len(dp.keys)))
            return list(dp.keys), rows, times
        if data.shape[0] != len(self.local_rows):
            raise ValueError(f"local_input: expected the {len(self.local_rows)} rows ShardedStream.local_rows, "
                             f"got {data.shape[0]}")
        sums = self.group_sums(data, group)
        x = np.concatenate([np.asarray(data, np.float32)] + [chmod.split_hi_lo(v) for v in sums], axis=0)
        if not len(starts):
            return list(dp.keys), np.empty((0, len(dp.keys))), times
        out, mask = dp.engine.process_batch(x, starts, want_nan_mask=True)
        mask_all = self._gather_mask(mask, len(self.channels), group)
        if mask_all.any() and mask_all.shape[1] != len(dp.ch_names_used):
            raise IndexError("boolean index did not match: NaN handling needs every channel used")
        rows = dp.postprocess_batch(out, mask_all if mask_all.any() else np.zeros((len(out), len(dp.ch_names_used)), bool),
                                    normalised=dp._norm_in_engine)
        return list(dp.keys), rows, times


class MultiDeviceProcessor:
    """Single-process form of the channel shard: one plan per device, one host thread per plan
    (ctypes drops the GIL for the duration of every libnmx call, so the launch sequences, the host <-> device
    copies and the waits of all devices overlap).  Same surface as ``DataProcessor`` for what ``Stream``
    uses: ``keys`` (reference order over ALL channels), ``process``, ``process_batch``, ``reset``.  The feature
    normaliser is per column, so each device normalises its own columns.

    ``local_input=True`` (default; SURVEY 8e "CAR hoisted before scatter"): a device is handed ONLY the input rows
    its own channel block taps (its channels, plus the few rows bipolar references name) and two rows per type
    group holding (hi, lo) of sum_{j in group} nan_to_num(x_j), formed ONCE on the host in float64 and shared by the devices --
    the host-to-device volume per device is C / N rows (+ group rows) instead of the whole recording (C5: 4096
    channels at 30 kS/s).  Channel tables whose re-reference rows are not "a few named channels and / or one group
    average" (processing/rereference.py:52-86 only builds such rows) fall back to the replicated form:
    ``local_input=False`` hands every device the whole recording and its rows of the folded matrix."""

    def __init__(self, sfreq, settings, channels, line_noise=None, devices=(0,), window=None, lib=None,
                 verbose: bool = False, resample_features_at_new_rate: bool = False, local_input: bool = True) -> None:
        from concurrent.futures import ThreadPoolExecutor

        self.settings = NMSettings.load(settings)
        self.channels = chmod.load_channels(channels)
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        names, _, _ = chmod.channel_info(self.channels)
        kw = dict(line_noise=line_noise, verbose=False, window=window, lib=lib,
                  resample_features_at_new_rate=resample_features_at_new_rate)
        layout = DataProcessor(sfreq, self.settings, self.channels, dry_run=True, **kw)
        self.keys = list(layout.keys)
        self.sfreq_raw = layout.sfreq_raw
        self.ch_names_used = layout.ch_names_used
        col = {k: i for i, k in enumerate(self.keys)}
        self.parts, self._cols = [], []
        self.local_input = bool(local_input)
        while True:
            try:
                for i, dev in enumerate(devices):
                    shard = channel_shard(len(names), len(devices), i)
                    if not len(shard):
                        continue   # more devices than channels
                    dp = DataProcessor(sfreq, self.settings, self.channels, device=dev, channel_subset=shard,
                                       staging_slot=i, local_inputs=self.local_input, **kw)
                    self.parts.append(dp)
                    self._cols.append(np.array([col[k] for k in dp.keys], dtype=np.int64))
                break
            except NotImplementedError:
                if not self.local_input:
                    raise
                for p in self.parts:   # rows without the taps + group-sum structure: replicated input
                    p.engine.close()
                self.parts, self._cols, self.local_input = [], [], False
        self.h2d_rows = [len(p.local_rows) + 2 * len(p.local_groups) if self.local_input else len(self.channels)
                         for p in self.parts]   # input rows every device receives (profiles/r04_host_boundary.json)
        if self.local_input:
            # every group sum once, whoever needs it
            self._groups, gid = [], {}
            self._part_groups = []
            for p in self.parts:
                ids = []
                for members in p.local_groups:
                    key = tuple(int(j) for j in members)
                    if key not in gid:
                        gid[key] = len(self._groups)
                        self._groups.append(np.asarray(key, dtype=np.int64))
                    ids.append(gid[key])
                self._part_groups.append(ids)
        self.devices = devices[:len(self.parts)]
        self.verbose = verbose
        self.settings_token = None
        self._norm_in_engine = all(p._norm_in_engine for p in self.parts)
        self._pool = ThreadPoolExecutor(max_workers=len(self.parts))
        # user-registered features see ALL channels (features/feature_processor.py:52-53): the parts hand back the
        # pre-processed windows of their channel blocks, the coordinator runs the plugins on the joined window
        from . import user_features as _registered

        self._user = None
        if _registered:
            self._user = UserColumns(self.settings, self.ch_names_used, self.sfreq_raw, self.keys,
                                     device=self.devices[0], lib=lib)
        self._user_chunk = 64
        # threads of libnmx's staging passes.  The library's default (an eighth of the machine, 4 - 16) assumes eight ranks
        # share a node; here ONE process feeds every device, and the pass that reads the float64 recording for all parts sets
        # the rate beyond two of them (profiles/r06_staging_bandwidth.json: 16 threads 50 - 70 GB/s of source, 64: 65 - 90;
        # a device consumes ~31 GB/s at 150 k hops/s): eight per device, at most 64 and half the machine.  NMX_HOST_THREADS
        # overrides.
        self.stage_threads = 0
        if len(self.devices) > 2 and not os.environ.get("NMX_HOST_THREADS"):
            self.stage_threads = int(max(16, min(64, 8 * len(self.devices), (os.cpu_count() or 32) // 2)))
        self.pipeline_min = (64, 1 << 20)   # hops, samples: below, staging and widening are not worth their threads

    @property
    def engine(self):
        return self.parts[0].engine   # window length / input shape are the same on every device

    @property
    def user_features(self) -> dict:
        return self._user.features if self._user is not None else {}

    @property
    def user_keys(self):
        return self._user.user_keys if self._user is not None else None

    def reset(self) -> None:
        for p in self.parts:
            p.reset()
        if self._user is not None:
            self._user.reset()

    def _merge(self, rows) -> np.ndarray:
        """The parts' tables side by side -> the global column order (a gather with one precomputed permutation, row
        blocks on the conversion threads: a per-part fancy-index scatter of 8 000 columns cost 30 ms per 1024 hops)."""
        from .engine import _pool

        n, F = rows[0].shape[0], len(self.keys)
        n_builtin = int(sum(len(c) for c in self._cols))
        if getattr(self, "_gather", None) is None or len(self._gather) != n_builtin:
            order = np.concatenate(self._cols)                 # global column of every concatenated column
            self._gather = np.argsort(order, kind="stable")     # concatenated column of every global built-in column
            self._gather_dst = np.sort(order)
        out = np.full((n, F), np.nan) if F != n_builtin else np.empty((n, F))
        cat = rows[0] if len(rows) == 1 else None
        edges = [(i * n) // 8 for i in range(9)] if n >= 64 else [0, n]
        contiguous = F == n_builtin or np.array_equal(self._gather_dst, np.arange(n_builtin))

        def job(i):
            a, b = edges[i], edges[i + 1]
            if a == b:
                return
            blk = cat[a:b] if cat is not None else np.concatenate([r[a:b] for r in rows], axis=1)
            if contiguous:
                np.take(blk, self._gather, axis=1, out=out[a:b, :n_builtin])
            else:
                out[a:b][:, self._gather_dst] = blk[:, self._gather]

        list(_pool().map(job, range(len(edges) - 1)))
        return out

    def _column_runs(self):
        """Per part: (first column in the table, first column of the part's row, length) of every contiguous run."""
        if getattr(self, "_runs", None) is None:
            self._runs = []
            for cols in self._cols:
                cut = np.flatnonzero(np.diff(cols) != 1) + 1
                first = np.concatenate([[0], cut]).astype(np.int64)
                length = np.diff(np.concatenate([first, [len(cols)]]))
                self._runs.append(np.ascontiguousarray(np.stack([cols[first], first, length], axis=1), dtype=np.int64)
                                  if len(cols) else np.zeros((0, 3), np.int64))
        return self._runs

    def _merge_widen(self, outs) -> np.ndarray:
        """The parts' float32 rows (engine order) -> the float64 table in the global column order, ONE pass
        (nmx_host_widen_rows): widening each part on its own and gathering the joined table afterwards wrote the
        82 MB of a 1024-hop table twice and read it once more.  The reference's keys run feature by feature, channel by
        channel: a part's columns land in a few dozen contiguous runs."""
        lib = self.parts[0].engine.lib
        n, F = outs[0].shape[0], len(self.keys)
        self._column_runs()
        n_builtin = int(sum(len(c) for c in self._cols))
        table = table_empty((n, F), np.nan if F != n_builtin else None)
        for o, runs in zip(outs, self._runs):
            o = o if (o.dtype == np.float32 and o.ndim == 2 and o.strides[1] == 4 and o.strides[0] % 4 == 0
                      and o.strides[0] > 0) else np.ascontiguousarray(o, dtype=np.float32)
            if n and len(runs):
                lib.check(lib.lib.nmx_host_widen_rows(table.ctypes.data, F, o.ctypes.data, o.strides[0] // 4, 0, n,
                                                      runs.ctypes.data, len(runs), 0))
        return table

    # -- local input: what every device is handed ------------------------------------------------------
    def _local_inputs(self, data: np.ndarray):
        """-> per part float32 [rows of the part + its group-sum rows, T] in the part's page-locked staging array; the
        group sums are formed once in float64 from nan_to_num(x) (the reference cleans before it re-references,
        stream/data_processor.py:255) by libnmx's staging helpers: one pass over the recording for the sums, one for the
        rows (the NumPy reduction along the channel axis ran at ~3 GB/s and set the rate of the whole stream)."""
        data = np.asarray(data)
        if not (data.dtype in (np.float32, np.float64) and data.strides[1] == data.itemsize and data.strides[0] > 0
                and data.strides[0] % data.itemsize == 0):
            data = np.ascontiguousarray(data, dtype=np.float32 if data.dtype == np.float32 else np.float64)
        lib = self.parts[0].engine.lib
        T, f64, ld = data.shape[1], int(data.dtype == np.float64), data.strides[0] // data.itemsize
        # (the float32-ROUNDED samples are summed: that is what a device that holds the rows itself adds up)
        hilo = []
        for g in self._groups:
            v, rows = np.empty(T), np.ascontiguousarray(g, dtype=np.int32)
            lib.check(lib.lib.nmx_host_group_sums(v.ctypes.data, data.ctypes.data, f64, ld, rows.ctypes.data, len(rows), 0, T, 0))
            hilo.append(chmod.split_hi_lo(v))
        xs = []
        for p, ids in zip(self.parts, self._part_groups):
            nl = len(p.local_rows)
            x = p.engine._pinned.array("x_local", (nl + 2 * len(ids), T), np.float32)
            rows = np.ascontiguousarray(p.local_rows, dtype=np.int32)
            lib.check(lib.lib.nmx_host_stage_rows(x.ctypes.data, x.strides[0] // 4, data.ctypes.data, f64, ld,
                                                  rows.ctypes.data, nl, 0, T, None, 0))
            for q, gi in enumerate(ids):
                x[nl + 2 * q:nl + 2 * q + 2] = hilo[gi]
            xs.append(x)
        return xs

    def _stage_local_slice(self, data, f64, ld, a, b, xs, plan):
        """Samples [a, b) of every part's staging array: its rows, and the hi / lo rows of its group sums -- the
        recording is read once (nmx_host_stage_parts: rows out and group sums block by block)."""
        lib = self.parts[0].engine.lib
        dst, gptr, grows, sums, sum_ptrs = plan
        lib.check(lib.lib.nmx_host_stage_parts(data.ctypes.data, f64, ld, data.shape[0], a, b, dst.ctypes.data, len(sums),
                                               gptr.ctypes.data, grows.ctypes.data, sum_ptrs.ctypes.data, self.stage_threads))
        hilo = [chmod.split_hi_lo(v[a:b]) for v in sums]
        for ids, rows, x in zip(self._part_groups, self._rows_i32, xs):
            nl = len(rows)
            for q, gi in enumerate(ids):
                x[nl + 2 * q:nl + 2 * q + 2, a:b] = hilo[gi]

    def _process_pipelined(self, data, starts, spare_cols: int = 0):
        """The whole batch with the coordinator's passes NEXT to the device work (HotPathEngine.run_pipelined): one
        thread stages the recording slice by slice for every part -- group sums and the part's rows with local input,
        one shared float32 copy otherwise -- and publishes its progress to every plan; every part widens its rows into
        its columns of the table as its chunks land."""
        import threading

        from .engine import parallel_cast

        data = np.asarray(data)
        if not (data.dtype in (np.float32, np.float64) and data.strides[1] == data.itemsize and data.strides[0] > 0
                and data.strides[0] % data.itemsize == 0):
            data = np.ascontiguousarray(data, dtype=np.float32 if data.dtype == np.float32 else np.float64)
        engines = [p.engine for p in self.parts]
        lib = engines[0].lib
        n, F, T = len(starts), len(self.keys), data.shape[1]
        f64, ld = int(data.dtype == np.float64), data.strides[0] // data.itemsize
        runs = self._column_runs()
        n_builtin = int(sum(len(c) for c in self._cols))
        table = table_empty((n, F + spare_cols), np.nan if F != n_builtin else None)
        if self.local_input:
            if getattr(self, "_rows_i32", None) is None:
                self._rows_i32 = [np.ascontiguousarray(p.local_rows, dtype=np.int32) for p in self.parts]
                self._groups_i32 = [np.ascontiguousarray(g, dtype=np.int32) for g in self._groups]
            xs = [e._pinned.array("x_local", (len(r) + 2 * len(ids), T), np.float32)
                  for e, r, ids in zip(engines, self._rows_i32, self._part_groups)]

            # where every source row goes (pointer to sample 0 of its destination row), the groups as CSR, their sums
            dst = np.zeros(data.shape[0], dtype=np.uint64)
            for rows, x in zip(self._rows_i32, xs):
                dst[rows] = x.ctypes.data + np.arange(len(rows), dtype=np.uint64) * np.uint64(x.strides[0])
            gptr = np.concatenate([[0], np.cumsum([len(g) for g in self._groups_i32])]).astype(np.int32)
            grows = (np.concatenate(self._groups_i32) if self._groups_i32 else np.zeros(0)).astype(np.int32)
            sums = [np.empty(T) for _ in self._groups_i32]
            sum_ptrs = np.array([v.ctypes.data for v in sums], dtype=np.uint64)
            plan = (dst, gptr, grows, sums, sum_ptrs)

            def stage(a, b):
                self._stage_local_slice(data, f64, ld, a, b, xs, plan)
        else:
            dcs = [e._host_offsets(data) for e in engines]   # (every plan decides from the same rows: the same constants)
            x = engines[0]._pinned.array("x_shared", data.shape, np.float32)
            xs = [x] * len(engines)

            def stage(a, b):
                parallel_cast(x[:, a:b], data[:, a:b], dcs[0], lib)
        ctrs = [e.pipeline_counters() for e in engines]
        failed: list = []

        def stage_all():
            try:
                edges = engines[0].pipeline_edges(starts, T)
                for a, b in zip(edges[:-1], edges[1:]):
                    stage(a, b)
                    for c in ctrs:
                        c[0] = b
            except BaseException as e:   # noqa: BLE001 -- reported below; the plans run out on what is there
                failed.append(e)
                for c in ctrs:
                    c[0] = T

        th = threading.Thread(target=stage_all, daemon=True)
        th.start()
        try:
            masks = list(self._pool.map(
                lambda k: engines[k].run_pipelined(xs[k], starts, table, runs[k], ctr=ctrs[k], want_nan_mask=True),
                range(len(engines))))
        finally:
            th.join()
        if failed:
            raise failed[0]
        mask = self._full_mask(masks, len(self.channels)) if self.local_input else masks[0]
        return table, mask

    def _full_mask(self, masks, n_all: int) -> np.ndarray:
        """NaN mask over ALL input rows from every part's local rows (a group-sum row is never NaN)."""
        full = np.zeros((masks[0].shape[0], n_all), dtype=bool)
        for p, m in zip(self.parts, masks):
            r, nl = p.local_rows, len(p.local_rows)
            if nl and list(r) == list(range(r[0], r[0] + nl)):
                full[:, r[0]:r[0] + nl] |= m[:, :nl]
            else:
                full[:, r] |= m[:, :nl]
        return full

    def _run_parts(self, data, starts, tapped: bool, staged: bool = False):
        """Every part on its own thread -> [(float32 rows, mask over ALL input rows, windows or None)].  ``staged``: the
        rows are views of the parts' page-locked output staging (valid until their next call)."""
        if not self.local_input:
            if tapped:
                return list(self._pool.map(lambda p: p.process_batch_tapped(data, starts), self.parts))
            return [(o, m, None) for o, m in self._pool.map(
                lambda p: p.engine.process_batch(data, starts, want_nan_mask=True, staged_output=staged), self.parts)]
        xs = self._local_inputs(data)

        W_in = self.parts[0].engine.W_in
        # nothing between the incoming rows and the features on any part (no re-reference, channel pick or filter): the
        # plugins get the exact float64 window, as the one-plan stream gives them (DataProcessor._host_windows)
        identity = all(p.engine.preprocessing_is_identity for p in self.parts)

        def job(px):
            p, x = px
            if tapped and identity:
                o, m = p.engine.process_batch(x, starts, want_nan_mask=True)
                rows = list(p.local_rows)   # (identity: exactly the part's channels, no group-sum rows)
                wins = np.stack([np.nan_to_num(np.asarray(data[rows, int(a):int(a) + W_in], dtype=np.float64)) for a in starts])
                return o, m, wins
            if tapped:
                o, m, pre = p.engine.process_batch(x, starts, want_nan_mask=True, tap=True)
                return o, m, pre.astype(np.float64)
            o, m = p.engine.process_batch(x, starts, want_nan_mask=True, staged_output=staged)
            return o, m, None

        got = list(self._pool.map(job, zip(self.parts, xs)))
        full = self._full_mask([m for _, m, _ in got], len(self.channels))
        return [(o, full, w) for o, _, w in got]

    def process_batch(self, data: np.ndarray, starts: np.ndarray, spare_cols: int = 0) -> np.ndarray:
        """``spare_cols``: as DataProcessor.process_batch (extra columns behind the features, pipelined path only)."""
        starts = np.asarray(starts, dtype=np.int64)
        if self._user is None and all(p.feature_normalizer is None and (p.device_normalizer is None or p._norm_in_engine)
                                      for p in self.parts):
            # nothing between the engines' rows and the table but the widening and the NaN policy
            if (len(starts) >= self.pipeline_min[0] and np.size(data) >= self.pipeline_min[1]
                    and os.environ.get("NMX_PIPELINE", "1") != "0"):
                table, mask = self._process_pipelined(data, starts, spare_cols)
            else:
                got = self._run_parts(data, starts, False, staged=True)
                table = self._merge_widen([o for o, _, _ in got])
                mask = got[0][1]   # over ALL incoming rows, the same for every part
            if mask.any():     # every key that contains the name of a channel whose window held a NaN := NaN (:297-306)
                if mask.shape[1] != len(self.ch_names_used):
                    raise IndexError("boolean index did not match: NaN handling needs every channel used")
                nan_cols = _LazyNanCols(self.keys, self.ch_names_used)
                for ci in np.where(mask.any(axis=0))[0]:
                    table[np.ix_(mask[:, ci], nan_cols[ci])] = np.nan
            return table
        if self._user is None:
            got = self._run_parts(data, starts, False)
            rows = [p.postprocess_batch(o, m, normalised=p._norm_in_engine) for p, (o, m, _) in zip(self.parts, got)]
            return self._merge(rows)
        tables = []
        for i in range(0, len(starts), self._user_chunk):
            st_ = starts[i:i + self._user_chunk]
            got = self._run_parts(data, st_, True)
            # contiguous channel blocks in device order: the joined window is the single-device one
            user = self._user.rows(np.concatenate([w for _, _, w in got], axis=1))
            builtin = [p.postprocess_batch(o, m, normalised=p._norm_in_engine) for p, (o, m, _) in zip(self.parts, got)]
            table = self._merge(builtin)
            table[:, self._user.cols] = user
            mask = got[0][1]   # replicated input: every part saw every incoming row
            if mask.any():     # the plugin keys follow the substring NaN policy too (stream/data_processor.py:297-306)
                nan_cols = _LazyNanCols(self.keys, self.ch_names_used)
                for ci in np.where(mask.any(axis=0))[0]:
                    table[np.ix_(mask[:, ci], nan_cols[ci])] = np.nan
            tables.append(table)
        return np.concatenate(tables) if tables else np.empty((0, len(self.keys)))

    # -- ragged window lengths: the protocol of DataProcessor.ragged_* over the parts --------------------------------
    def ragged_prepare(self) -> None:
        for p in self.parts:
            p.ragged_prepare()
        self._norm_in_engine = False

    def ragged_state(self):
        st = [p.engine.export_state() for p in self.parts]
        return st if any(st) else None

    def ragged_set_state(self, state) -> None:
        for p, s_ in zip(self.parts, state):
            if s_:
                p.engine.import_state(s_)

    def ragged_run(self, data: np.ndarray, starts: np.ndarray):
        got = self._run_parts(data, np.asarray(starts, dtype=np.int64), self._user is not None)
        wins = None
        if self._user is not None:   # contiguous channel blocks in device order: the joined window is the single-device one
            joined = np.concatenate([w for _, _, w in got], axis=1)
            wins = [joined[j] for j in range(joined.shape[0])]
        return [o for o, _, _ in got], got[0][1], wins

    def ragged_finish(self, runs) -> np.ndarray:
        masks = np.concatenate([r[1] for r in runs])
        zero = np.zeros_like(masks)
        # every part normalises its own columns over ALL hops (the normaliser is per column)
        rows = [p.postprocess_batch(np.concatenate([r[0][i] for r in runs]), zero, normalised=False)
                for i, p in enumerate(self.parts)]
        table = self._merge(rows)
        if self._user is not None:
            user = self._user.rows([w for r in runs for w in r[2]])
            if table.shape[1] != len(self.keys):   # the first call appended the plugin keys
                wide = np.full((table.shape[0], len(self.keys)), np.nan)
                wide[:, :table.shape[1]] = table
                table = wide
            table[:, self._user.cols] = user
        if masks.any():
            nan_cols = _LazyNanCols(self.keys, self.ch_names_used)
            if masks.shape[1] != len(self.ch_names_used):
                raise IndexError("boolean index did not match: NaN handling needs every channel used")
            for ci in np.where(masks.any(axis=0))[0]:
                table[np.ix_(masks[:, ci], nan_cols[ci])] = np.nan
        return table

    def process(self, data: np.ndarray) -> dict:
        if self._user is not None or self.local_input:
            row = self.process_batch(np.asarray(data), np.zeros(1, np.int64))[0]
            return dict(zip(self.keys, row.tolist()))
        parts = list(self._pool.map(lambda p: p.process(data), self.parts))
        merged = {}
        for d in parts:
            merged.update(d)
        return {k: merged[k] for k in self.keys}

    def close(self) -> None:
        for p in self.parts:
            p.engine.close()
        self._pool.shutdown(wait=False)


def merge_shards(all_keys: list[str], shards) -> np.ndarray:
    """Assemble shard outputs [(keys, rows), ...] into the global column order."""
    n = shards[0][1].shape[0]
    out = np.full((n, len(all_keys)), np.nan)
    col = {k: i for i, k in enumerate(all_keys)}
    for keys, rows in shards:
        out[:, [col[k] for k in keys]] = rows
    return out


def gather_dataframe(local_keys, local_rows, times, all_keys, group=None):
    """Rank 0 returns the global feature DataFrame (others None).  Needs an initialised
    torch.distributed process group when world_size > 1."""
    import pandas as pd
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        parts = [(local_keys, local_rows)]
        rank = 0
    else:
        rank = dist.get_rank(group)
        gathered = [None] * dist.get_world_size(group) if rank == 0 else None
        dist.gather_object((local_keys, local_rows), gathered, dst=0, group=group)
        parts = gathered
    if rank != 0:
        return None
    df = pd.DataFrame(merge_shards(all_keys, parts), columns=all_keys)
    df["time"] = times
    return df

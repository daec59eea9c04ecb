"""Host side of the hot path: settings -> flat C plan -> HIP kernels (via libnmx.so).

``HotPathEngine`` turns a (duck-typed) ``NMSettings`` + channel names into an
``nmx_plan_desc`` -- band -> FFT-bin tables, FIR taps designed on the host, the output column
of every (feature, channel) in the reference's key order -- and drives the C ABI:

    process_window(data[C_in, W] float64) -> float32[F]      one hop, the reference call shape
    process_batch(data[C_in, T], starts)   -> float32[n, F]   many hops resident on the GPU

Key order reproduced: features in ``FeatureSelector`` field order
(stream/settings.py:41-55 via utils/types.py:135-140), and inside each feature the nesting of
the reference class (hjorth_raw.py:37-40, bandpower.py:131-145, oscillatory.py:102-117,
sharpwaves.py:169-197,302-326, bursts.py:119-125,262-298, linelength.py:17-19).
"""

from __future__ import annotations

import ctypes as C
import math
import os
import time
from collections.abc import Sequence

import numpy as np

from . import _lib, fir_design
from ._lib import Cols, PlanDesc
from .settings import validation_error

_FEATURE_BITS = {"raw_hjorth": _lib.F_HJORTH, "return_raw": _lib.F_RAW,
                 "bandpass_filter": _lib.F_BANDPOWER, "stft": _lib.F_STFT, "fft": _lib.F_FFT,
                 "welch": _lib.F_WELCH, "sharpwave_analysis": _lib.F_SHARPWAVE,
                 "bursts": _lib.F_BURSTS, "linelength": _lib.F_LINELENGTH}
_OUT_OF_SCOPE = {"fooof", "nolds", "coherence", "mne_connectivity", "bispectrum"}
_BURST_SLOTS = [("duration", ["duration_mean", "duration_max"]),
                ("amplitude", ["amplitude_mean", "amplitude_max"]),
                ("burst_rate_per_s", ["burst_rate_per_s"]), ("in_burst", ["in_burst"])]


def _enabled(sel) -> list[str]:
    return list(sel.get_enabled())


def _cols(base=0, ch=0, a=0, b=0) -> Cols:
    return Cols(int(base), int(ch), int(a), int(b))


_POOL = None


def _pool():
    """A few host threads for the dtype conversions around a batch (NumPy's casting loops release the GIL):
    float64 recording -> float32 staging and float32 features -> float64 table are 100+ MB each."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)))
    return _POOL


def _rows_ok(a: np.ndarray, dtype) -> bool:
    return (a.ndim == 2 and a.dtype == dtype and a.strides[1] == a.itemsize and a.strides[0] > 0
            and a.strides[0] % a.itemsize == 0)


def _ld(a: np.ndarray) -> int:
    """Row pitch of a row-contiguous 2-D array in elements.  (NumPy is free to report ANY stride along an axis of
    length one -- the transpose of a (samples, 1) array has a row "pitch" of one element: a single row's pitch is its
    length.)"""
    return a.strides[0] // a.itemsize if a.shape[0] > 1 else max(a.shape[1], 1)


def parallel_cast(dst: np.ndarray, src: np.ndarray, sub: np.ndarray | None = None, lib=None) -> None:
    """dst[...] = src (with cast), first axis split over the conversion threads.  ``sub``: one float64 constant per row,
    subtracted BEFORE the cast (the engine's offset split: the cast then rounds at the signal's magnitude).  With the
    loaded library and row-contiguous float arrays the pass runs in libnmx's staging helpers (nmx_host_stage_rows /
    nmx_host_widen_rows: one pass at memory speed; NumPy's buffered casts manage 1 - 3 GB/s) -- same values."""
    n = dst.shape[0]
    if lib is not None and dst.ndim == 2 and dst.shape == getattr(src, "shape", None) and dst.size:
        if _rows_ok(dst, np.float32) and (_rows_ok(src, np.float32) or _rows_ok(src, np.float64)):
            k = None if sub is None else np.ascontiguousarray(sub, dtype=np.float64)
            lib.check(lib.lib.nmx_host_stage_rows(dst.ctypes.data, dst.strides[0] // 4, src.ctypes.data,
                                                  int(src.dtype == np.float64), src.strides[0] // src.itemsize, None, n, 0,
                                                  dst.shape[1], None if k is None else k.ctypes.data, 0))
            return
        if sub is None and _rows_ok(dst, np.float64) and _rows_ok(src, np.float32):
            runs = np.array([0, 0, dst.shape[1]], dtype=np.int64)
            lib.check(lib.lib.nmx_host_widen_rows(dst.ctypes.data, dst.strides[0] // 8, src.ctypes.data, src.strides[0] // 4,
                                                  0, n, runs.ctypes.data, 1, 0))
            return

    def put(a, b):
        if sub is None:
            dst[a:b] = src[a:b]
        else:
            dst[a:b] = np.asarray(src[a:b], dtype=np.float64) - sub[a:b, None]

    if dst.size < (1 << 20) or n < 2:
        put(0, n)
        return
    k = min(n, 8)
    edges = [(i * n) // k for i in range(k + 1)]
    list(_pool().map(lambda i: put(edges[i], edges[i + 1]), range(k)))


class _TableLease:
    """Owner of a result table's memory: hands its buffer back to the pool when the last array on it is gone."""

    def __init__(self, raw: np.ndarray, pool: "_TablePool") -> None:
        self._raw, self._pool = raw, pool
        self.__array_interface__ = raw.__array_interface__

    def __del__(self):
        try:
            self._pool._give_back(self._raw)
        except Exception:   # (interpreter shutdown)
            pass


class _TablePool:
    """Recycled float64 result tables.  A stream of the reference's shape hands a fresh [hops, features] float64 table to
    its caller per run (stream/stream.py:319-343): 75 - 80 MB whose first touch is ~20 000 page faults and whose release
    another ~3 ms of munmap (measured on the GPU box: touch 4 - 85 ms depending on what the kernel finds for its huge
    pages, free 2.8 - 3.2 ms), per run, on the caller's thread.  The tables here are ordinary ndarrays whose memory goes
    back to this pool instead of the system when the caller drops them (DataFrame included) -- the next run finds its pages
    mapped.  At most ``keep`` idle buffers are held (``release_tables()`` frees them); small tables are not pooled."""

    MIN_BYTES = 1 << 22

    def __init__(self, keep: int = 2) -> None:
        import collections
        import threading

        self.keep, self._free, self._lock = keep, [], threading.Lock()
        # buffers coming home from `_TableLease.__del__`: a finaliser can run on ANY allocation of the thread that holds
        # the lock (a cyclic-GC pass inside `empty`), so it takes no lock -- deque.append is atomic -- and the list is
        # merged by the next call that does
        self._returned = collections.deque()

    def _merge(self) -> None:   # (lock held)
        while self._returned:
            self._free.append(self._returned.popleft())
        del self._free[:max(0, len(self._free) - self.keep)]

    def empty(self, shape, fill=None) -> np.ndarray:
        n = int(np.prod(shape))
        if n * 8 < self.MIN_BYTES:
            return np.empty(shape) if fill is None else np.full(shape, fill)
        raw = None
        with self._lock:
            self._merge()
            for i in range(len(self._free)):
                if n <= self._free[i].size <= n + n // 4:
                    raw = self._free.pop(i)
                    break
        if raw is None:
            raw = np.empty(n)
        arr = np.asarray(_TableLease(raw, self))[:n].reshape(shape)
        if fill is not None:
            arr.fill(fill)
        return arr

    def _give_back(self, raw: np.ndarray) -> None:
        self._returned.append(raw)
        if len(self._returned) > 2 * self.keep + 2 and self._lock.acquire(blocking=False):   # (nobody allocates: trim here)
            try:
                self._merge()
            finally:
                self._lock.release()

    def clear(self) -> int:
        with self._lock:
            self._merge()
            n = len(self._free)
            self._free.clear()
        return n


_TABLES = _TablePool()


def table_empty(shape, fill=None) -> np.ndarray:
    """A float64 result table from the recycling pool (``_TablePool``)."""
    return _TABLES.empty(shape, fill)


def release_tables() -> int:
    """Free the idle result-table buffers of the pool; returns how many."""
    return _TABLES.clear()


_STAGING: dict = {}


def _staging(lib, device: int = 0, slot: int = 0) -> "_Pinned":
    """ONE staging pool per loaded library, shared by every engine (page-locking ~170 MB costs tens of
    milliseconds -- Stream.run builds a fresh engine per run, like the reference builds a fresh
    DataProcessor).  Engines are not re-entrant (include/nmx.h), neither is the pool -- one pool per device AND
    ``slot``: the engines of a multi-device stream run on one host thread each and take the slot of their part index,
    so two parts never share a staging array even when they name the same device (``devices=[0, 0]``).  Pools are
    reference counted and cached (see ``_release_staging`` / ``release_staging``)."""
    key = (str(lib.path), int(device), int(slot))
    if key not in _STAGING:
        _STAGING[key] = _Pinned(lib, key)
    _STAGING[key].users += 1
    return _STAGING[key]


def _release_staging(pool: "_Pinned") -> None:
    """An engine closed.  The pool stays cached: page-locking its ~170 MB again costs ~17 ms per Stream (measured:
    8 ms hipHostMalloc + 9 ms hipHostFree against a 29 ms Stream.run), and its size is bounded by the largest batch
    ever staged (two named arrays, grown on demand).  ``release_staging()`` frees the pools nobody uses; every pool is
    freed at interpreter exit."""
    pool.users -= 1


def release_staging() -> int:
    """Free the page-locked staging pools that no open engine uses -- and hand the device memory libnmx keeps of destroyed
    plans back to the driver (nmx_device_pool_trim: a co-tenant of the GPU cannot reclaim it); returns how many staging
    pools were freed."""
    n = 0
    libs = {}
    for key in [k for k, p in _STAGING.items() if p.users <= 0]:
        pool = _STAGING.pop(key)
        libs[id(pool.lib)] = pool.lib
        pool.close()
        n += 1
    for p in _STAGING.values():
        libs[id(p.lib)] = p.lib
    for lib in libs.values():
        trim_device_pool(lib)
    return n


def trim_device_pool(lib=None, keep_bytes: int = 0) -> int:
    """Idle device blocks of destroyed plans back to the driver until at most ``keep_bytes`` stay cached -> bytes freed."""
    lib = lib if lib is not None else _lib.get_library()
    freed = C.c_int64(0)
    lib.check(lib.lib.nmx_device_pool_trim(int(keep_bytes), C.byref(freed)))
    return int(freed.value)


def _free_all_staging() -> None:   # atexit
    for key in list(_STAGING):
        try:
            pool = _STAGING.pop(key)
            pool.close()
            pool.lib.lib.nmx_device_pool_trim(0, None)
        except Exception:
            pass


import atexit  # noqa: E402

atexit.register(_free_all_staging)


_scratch_tls = None


def _offset_scratch(rows: int, cols: int) -> np.ndarray:
    """A C-ordered float64 [rows, cols] scratch array of the calling thread (`HotPathEngine._host_offsets`)."""
    global _scratch_tls
    if _scratch_tls is None:
        import threading

        _scratch_tls = threading.local()
    buf = getattr(_scratch_tls, "buf", None)
    if buf is None or buf.size < rows * cols:
        buf = _scratch_tls.buf = np.empty(max(rows * cols, 1 << 16), np.float64)
    return buf[:rows * cols].reshape(rows, cols)


class _Pinned:
    """Page-locked staging arrays (nmx_host_alloc), grown on demand and reused across calls."""

    def __init__(self, lib, key=None) -> None:
        self.lib, self._bufs, self.users, self.key = lib, {}, 0, key

    def array(self, name: str, shape, dtype) -> np.ndarray:
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr, cap = self._bufs.get(name, (None, 0))
        if cap < n:
            if ptr:
                self.lib.lib.nmx_host_free(ptr)
            p = C.c_void_p()
            self.lib.check(self.lib.lib.nmx_host_alloc(n + n // 8 + 64, C.byref(p)))
            ptr, cap = p.value, n + n // 8 + 64
            self._bufs[name] = (ptr, cap)
        buf = (C.c_char * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self) -> None:
        for ptr, _ in self._bufs.values():
            if ptr:
                self.lib.lib.nmx_host_free(ptr)
        self._bufs = {}


MAX_PLAN_WINDOW = 16384   # nmx_plan_create: window in [4, 16384] samples


def long_segments(n_samples: int, halo: int, window: int = MAX_PLAN_WINDOW):
    """A FIR filter over a recording longer than one plan's window, for the stand-alone filter classes (MNEFilter,
    NotchFilter, PreprocessingFilter: the reference's filter any length): windows ``[lo, lo + window)`` of the recording
    and the output samples ``[a, b)`` each of them is exact for -- every sample at least ``halo`` (half the filter, or
    the sum of the halves of a chain) away from a cut that is not an end of the recording, so that it sees the same
    input samples as in one long convolution; at the two ends of the recording the window's own edge handling (zeros,
    or the notch's reflection) IS the recording's.  Yields (lo, a, b)."""
    step = window - 2 * halo
    if step < 1:
        raise ValueError(f"a filter chain of {2 * halo + 1} taps leaves no room for data in one {window}-sample window")
    a = 0
    while a < n_samples:
        lo = min(max(a - halo, 0), n_samples - window)
        b = n_samples if lo + window >= n_samples else lo + window - halo
        yield lo, a, b
        a = b


class HotPathEngine:
    """One plan on one GPU for ``len(ch_names)`` channels."""

    def __init__(self, settings, ch_names: Sequence[str], sfreq: float, *,
                 features: Sequence[str] | None = None, ref_matrix: np.ndarray | None = None,
                 notch_taps: np.ndarray | None = None, device: int = 0,
                 lib: _lib.NmxLibrary | None = None, bank_taps: np.ndarray | None = None,
                 sharpwave_taps: Sequence[np.ndarray] | None = None,
                 window: int | None = None, dry_run: bool = False,
                 resample_from: float | None = None, raw_window: int | None = None,
                 pre_taps: Sequence[np.ndarray] | None = None,
                 raw_norm: tuple | None = None, resample_to: float | None = None, staging_slot: int = 0) -> None:
        """``sfreq`` is the rate every feature and filter is DESIGNED with (what the reference passes to the
        feature constructors).  ``resample_from`` = sampling rate of the incoming windows when they are
        resampled (raw_resampling, processing/resample.py:19-60): incoming windows then hold ``raw_window``
        samples (default int(segment_length_features_ms / 1000 * resample_from)) and are resampled on the
        device to ``window`` = round(ratio * raw_window) samples, ratio = ``resample_to`` / resample_from.
        ``resample_to`` defaults to ``sfreq`` (features designed for the rate they see).  The reference
        instead keeps designing with the RAW rate (stream/data_processor.py:55,68,80): that is
        ``sfreq = resample_from`` with ``resample_to`` = the new rate -- windows of round(ratio * raw_window)
        samples analysed as if sampled at the raw rate (FFT / Welch segments longer than the window shrink
        to it as NumPy slicing / scipy.signal.welch do, band-pass tails are clamped, bursts keep
        seg_s = segment_length_features_ms / 1000).

        ``dry_run=True`` only derives the plan description and the key list (no library, no
        GPU) -- used to lay out the global column order when channels are sharded over GPUs."""
        self.lib = None if dry_run else (lib if lib is not None else _lib.get_library())
        self.settings = settings
        self.ch_names = list(ch_names)
        self.sfreq = float(sfreq)
        self.C = len(self.ch_names)
        enabled = _enabled(settings.features) if features is None else list(features)
        bad = [f for f in enabled if f in _OUT_OF_SCOPE or f not in _FEATURE_BITS]
        if bad:
            raise NotImplementedError(
                f"features {bad} are outside the accelerated hot path (SURVEY.md section 2)")
        self.enabled = enabled
        # window samples as the generator cuts them (stream/generator.py:34-53)
        self.W = int(window) if window is not None else int(settings.segment_length_features_ms / 1000 * sfreq)
        self.W_in, self.resample_ratio = self.W, 0.0
        target = self.sfreq if resample_to is None else float(resample_to)
        if resample_from is not None and float(resample_from) != target:
            self.resample_ratio = target / float(resample_from)
            self.W_in = (int(raw_window) if raw_window is not None
                         else int(settings.segment_length_features_ms / 1000 * float(resample_from)))
            if window is None:
                self.W = int(round(self.resample_ratio * self.W_in))   # mne.filter.resample: final_len
        self._keep: list = []   # arrays referenced by the C struct
        self._raw_norm = raw_norm   # (method "mean" | "zscore", clip, N samples, add samples)
        self._pre_taps = [np.asarray(t, dtype=np.float64) for t in (pre_taps or [])]
        if len(self._pre_taps) > 4:
            raise ValueError("at most 4 preprocessing_filter stages")
        self.keys: list[str] = []
        self.desc = self._build(ref_matrix, notch_taps, device, bank_taps, sharpwave_taps)
        self.n_outputs = len(self.keys)
        self.C_in = int(self.desc.n_channels_in)
        self._plan = C.c_void_p()
        self._dc = None        # host offsets in force (None: not decided yet -- `_host_offsets`)
        self._dc_any = False
        self._pinned = None
        if not dry_run:
            self.lib.check(self.lib.lib.nmx_plan_create(C.byref(self.desc), C.byref(self._plan)))
            self._pinned = _staging(self.lib, device, staging_slot)

    # ------------------------------------------------------------------------------------
    def _dptr(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._keep.append(arr)
        return arr.ctypes.data_as(C.POINTER(C.c_double))

    def _osc(self, osc_settings, name: str, n: int, freqs: np.ndarray, inclusive: bool,
             bands, base: int) -> tuple[_lib.OscDesc, int]:
        s = osc_settings
        if not s.windowlength_ms <= self.settings.segment_length_features_ms:
            raise AssertionError(
                f"oscillatory feature windowlength_ms = ({s.windowlength_ms}) needs to be smaller "
                f"than settings['segment_length_features_ms'] = "
                f"{self.settings.segment_length_features_ms}")
        o = _lib.OscDesc()
        # a segment longer than the window shrinks to it: x[:, -N:] (oscillatory.py:95) and the nperseg
        # clamp of scipy.signal.welch / stft; the band bins keep the grid of the NOMINAL length
        n_eff = min(int(n), self.W)
        o.n = n_eff
        o.log_transform = int(bool(s.log_transform))
        ests = _enabled(s.features)
        o.estimators = sum(_lib.EST_BITS[e] for e in ests)
        o.return_spectrum = int(bool(s.return_spectrum))
        for b, (_, (lo, hi)) in enumerate(bands):
            idx = np.where((freqs >= lo) & ((freqs <= hi) if inclusive else (freqs < hi)))[0]
            if idx.size and not np.array_equal(idx, np.arange(idx[0], idx[-1] + 1)):
                raise ValueError("non-contiguous band bins")
            if not idx.size and "max" in ests:   # np.max over an empty band (oscillatory.py:170)
                raise ValueError("zero-size array to reduction operation maximum which has no identity "
                                 f"({name}: band {bands[b][0]} holds no bin of the {len(freqs)}-bin grid)")
            o.bin_lo[b] = int(idx[0]) if idx.size else 0
            o.bin_hi[b] = int(idx[-1]) + 1 if idx.size else 0
            if o.bin_hi[b] > n_eff // 2 + 1:
                raise IndexError(f"{name}: band {bands[b][0]} reads bin {o.bin_hi[b] - 1} of a spectrum with "
                                 f"{n_eff // 2 + 1} bins (window shorter than the nominal segment)")
        # keys: band, estimator, channel  (oscillatory.py:102-112)
        o.cols = _cols(base, 1, len(ests) * self.C, self.C)
        for bname, _ in bands:
            for e in ests:
                self.keys += [f"{ch}_{name}_{bname}_{e}" for ch in self.ch_names]
        used = len(bands) * len(ests) * self.C
        if o.return_spectrum:
            if len(freqs) != n_eff // 2 + 1:
                raise IndexError(f"{name}: return_spectrum needs a window as long as the nominal segment")
            ints = [int(f) for f in freqs]
            if len(set(ints)) != len(ints):
                raise NotImplementedError(
                    f"{name}: return_spectrum with a bin spacing below 1 Hz yields colliding "
                    "'psd_<int(f)>' keys in the reference; not supported")
            o.psd_cols = _cols(base + used, len(freqs), 1, 0)
            for ch in self.ch_names:
                self.keys += [f"{ch}_{name}_psd_{i}" for i in ints]
            used += self.C * len(freqs)
        return o, used

    def _build(self, ref_matrix, notch_taps, device, bank_taps, sharpwave_taps) -> PlanDesc:
        st, C_, W, sfreq = self.settings, self.C, self.W, self.sfreq
        d = PlanDesc()
        d.abi_version = _lib.NMX_ABI_VERSION
        d.device = int(device)
        d.n_channels = C_
        d.window = W
        if self.resample_ratio:
            d.raw_window = int(self.W_in)
            d.resample_ratio = float(self.resample_ratio)
        if self._raw_norm is not None:              # raw_normalization, last pre-processor
            method, clip, n_hist, add = self._raw_norm
            # "quantile": scikit-learn's QuantileTransformer subsamples histories of more than 10 000 samples at random
            # (random_state=None: the reference is not reproducible there); the device draws its own uniformly
            # random subset per hop and channel (nmx_k_rawnorm.h) -- equal in distribution, exact below 10 000 samples
            codes = {"mean": 1, "zscore": 2, "median": 3, "zscore-median": 4, "robust": 5, "minmax": 6,
                     "quantile": 7, "power": 8}
            if method not in codes:
                raise NotImplementedError(f"raw_normalization method {method!r} has no device implementation")
            d.raw_norm_method = codes[method]
            d.raw_norm_clip = float(clip) if clip else 0.0
            d.raw_norm_n, d.raw_norm_add = int(n_hist), int(add)
        d.n_pre_filters = len(self._pre_taps)       # preprocessing_filter stages, before the notch
        for i, t in enumerate(self._pre_taps):
            d.pre_taps[i] = self._dptr(t)
            d.n_pre_taps[i] = len(t)
        d.sfreq = sfreq
        d.segment_length_s = float(st.segment_length_features_ms) / 1000.0
        d.feat_hz = float(st.sampling_rate_features_hz)
        bands = [(name, (float(fr[0]), float(fr[1]))) for name, fr in st.frequency_ranges_hz.items()]
        if len(bands) > _lib.NMX_MAX_BANDS:
            raise ValueError(f"at most {_lib.NMX_MAX_BANDS} frequency bands are supported")
        d.n_bands = len(bands)
        band_index = {name: i for i, (name, _) in enumerate(bands)}
        filters: list[dict] = []
        band_filter: dict[str, int] = {}

        def bank_filter(name: str) -> dict:
            """FIR of band `name` (shared by BandPower and Bursts: same MNEFilter arguments)."""
            if name not in band_filter:
                if bank_taps is not None:
                    taps = np.asarray(bank_taps)[band_index[name]]
                else:
                    taps = fir_design.band_pass_bank([bands[band_index[name]][1]], sfreq)[0]
                band_filter[name] = len(filters)
                filters.append({"taps": taps, "bp_seglen": 0, "bp_band": 0, "burst": -1, "sw": -1})
            return filters[band_filter[name]]

        feats = 0
        col = 0
        for f in self.enabled:
            feats |= _FEATURE_BITS[f]
            if f == "raw_hjorth":
                d.hjorth_cols = _cols(col, 3, 1, 0)
                for ch in self.ch_names:
                    self.keys += [f"{ch}_RawHjorth_Activity", f"{ch}_RawHjorth_Mobility",
                                  f"{ch}_RawHjorth_Complexity"]
                col += 3 * C_
            elif f == "return_raw":
                d.raw_cols = _cols(col, 1)
                self.keys += [f"{ch}_raw" for ch in self.ch_names]
                col += C_
            elif f == "linelength":
                d.linelength_cols = _cols(col, 1)
                self.keys += [f"{ch}_LineLength" for ch in self.ch_names]
                col += C_
            elif f == "fft":
                n = int(np.floor(st.fft_settings.windowlength_ms / 1000 * sfreq))
                freqs = np.fft.rfftfreq(n, 1 / np.floor(int(sfreq)))
                d.fft, used = self._osc(st.fft_settings, "fft", n, freqs, False, bands, col)
                col += used
            elif f == "welch":
                n = int(sfreq)
                freqs = np.fft.rfftfreq(n, 1 / n)
                d.welch, used = self._osc(st.welch_settings, "welch", n, freqs, False, bands, col)
                col += used
            elif f == "stft":
                n = int(st.stft_settings.windowlength_ms)   # ms used as samples (oscillatory.py:199)
                freqs = np.fft.rfftfreq(n, 1 / int(sfreq))
                d.stft, used = self._osc(st.stft_settings, "stft", n, freqs, True, bands, col)
                col += used
            elif f == "bandpass_filter":
                bp = st.bandpass_filter_settings
                bfeats = _enabled(bp.bandpower_features)
                if getattr(bp, "kalman_filter", False) and "activity" in bfeats:
                    # bandpower.py:125-126,147-156: one filter per channel for every band of
                    # kalman_filter_settings.frequency_bands (bands missing from
                    # frequency_ranges_hz never match a feature name and stay unused)
                    ks = st.kalman_filter_settings
                    d.bp_kalman_mask = sum(1 << band_index[b] for b in ks.frequency_bands if b in band_index)
                    d.kalman_Tp, d.kalman_sigma_w, d.kalman_sigma_v = float(ks.Tp), float(ks.sigma_w), float(ks.sigma_v)
                d.bp_features = sum(1 << ["activity", "mobility", "complexity"].index(x) for x in bfeats)
                d.bp_log_transform = int(bool(bp.log_transform))
                nf = len(bfeats)
                d.bp_cols = _cols(col, len(bands) * nf, nf, 1)
                for name, _ in bands:
                    fd = bank_filter(name)
                    fd["bp_seglen"] = min(int(np.floor(sfreq / 1000 * bp.segment_lengths_ms[name])), W)   # y[..., -seglen:]
                    fd["bp_band"] = band_index[name]
                for ch in self.ch_names:
                    for name, _ in bands:
                        self.keys += [f"{ch}_bandpass_{x}_{name}" for x in bfeats]
                col += C_ * len(bands) * nf
            elif f == "bursts":
                bs = st.bursts_settings
                names = list(bs.frequency_bands)
                for nme in names:
                    if nme not in band_index:
                        raise validation_error(f"bursting {nme} needs to be defined in "
                                               "settings['frequency_ranges_hz']",
                                               ["burst_settings", "frequency_bands"])
                groups = _enabled(bs.burst_features)
                slots, mask, bit = [], 0, 0
                for g, outs in _BURST_SLOTS:
                    for o in outs:
                        if g in groups:
                            mask |= 1 << bit
                        bit += 1
                # the reference emits the groups in burst_features order, which is _BURST_SLOTS order
                for g in groups:
                    slots += dict(_BURST_SLOTS)[g]
                d.n_burst_bands = len(names)
                d.burst_threshold = float(bs.threshold)
                d.burst_time_duration_s = float(bs.time_duration_s)
                d.burst_out_mask = mask
                d.burst_cols = _cols(col, len(names) * len(slots), len(slots), 1)
                for i, nme in enumerate(names):
                    bank_filter(nme)["burst"] = i
                for ch in self.ch_names:
                    for nme in names:
                        self.keys += [f"{ch}_bursts_{nme}_{s}" for s in slots]
                col += C_ * len(names) * len(slots)
            elif f == "sharpwave_analysis":
                col = self._sharpwave(d, filters, col, sharpwave_taps)
        d.features = feats
        d.n_outputs = col
        if len(filters) > _lib.NMX_MAX_FILTERS:
            raise ValueError(f"at most {_lib.NMX_MAX_FILTERS} FIR filters per plan are supported")
        d.n_filters = len(filters)
        for i, fd in enumerate(filters):
            taps = np.asarray(fd["taps"], np.float64)
            d.filters[i].taps = self._dptr(taps)
            d.filters[i].n_taps = len(taps)
            d.filters[i].bp_seglen = fd["bp_seglen"]
            d.filters[i].bp_band_index = fd["bp_band"]
            d.filters[i].burst_index = fd["burst"]
            d.filters[i].sw_index = fd["sw"]
        self.filter_taps = [np.asarray(fd["taps"], np.float64) for fd in filters]
        if notch_taps is not None:
            nt = np.asarray(notch_taps, np.float64)
            d.notch_taps = self._dptr(nt)
            d.n_notch_taps = len(nt)
        if ref_matrix is not None:
            R = np.ascontiguousarray(ref_matrix, np.float64)
            if R.ndim != 2 or R.shape[0] != C_:
                raise ValueError("ref_matrix must be [n_channels, n_channels_in]")
            d.ref_matrix = self._dptr(R)
            d.n_channels_in = R.shape[1]
        else:
            d.n_channels_in = C_
        assert len(self.keys) == col
        return d

    def _sharpwave(self, d: PlanDesc, filters: list, col: int, sharpwave_taps) -> int:
        st, sfreq, C_ = self.settings, self.sfreq, self.C
        sw = st.sharpwave_analysis_settings
        used = [f for f in _lib.SW_FEATURES if getattr(sw.sharpwave_features, f)]
        est_of = {f: [e for e in _lib.SW_ESTIMATORS if f in sw.estimator[e]] for f in used}
        for f in used:
            assert est_of[f], f"Add estimator key for {f}"
        combos = [(f, e) for f in used for e in est_of[f]]
        if len(combos) > _lib.NMX_MAX_SW_COMBOS:
            raise ValueError("too many sharp-wave (feature, estimator) pairs")
        names = []
        for i, fr in enumerate(sw.filter_ranges_hz):
            assert fr[1] < sfreq, ("Filter range has to be smaller than sfreq, "
                                   f"got sfreq {sfreq} and filter range {fr}")
            names.append(f"range_{fr[0]:.0f}_{fr[1]:.0f}")
            taps = (np.asarray(sharpwave_taps[i]) if sharpwave_taps is not None
                    else fir_design.band_pass(sfreq, fr[0], fr[1]))
            filters.append({"taps": taps, "bp_seglen": 0, "bp_band": 0, "burst": -1, "sw": i})
        d.n_sw_filters = len(names)
        d.sw_n_combos = len(combos)
        for i, (f, e) in enumerate(combos):
            d.sw_combo_feature[i] = _lib.SW_FEATURES.index(f)
            d.sw_combo_estimator[i] = _lib.SW_ESTIMATORS.index(e)
        d.sw_distance_peaks = float(sw.detect_troughs.distance_peaks_ms)      # sharpwaves.py:339-344
        d.sw_distance_troughs = float(sw.detect_troughs.distance_troughs_ms)
        d.sw_estimate_peaks = int(bool(sw.detect_peaks.estimate))
        d.sw_estimate_troughs = int(bool(sw.detect_troughs.estimate))
        between = bool(sw.apply_estimator_between_peaks_and_troughs)
        d.sw_between = int(between)
        NF = len(names)
        if between:
            reg = [(f, e) for f, e in combos if f != "num_peaks"]
            d.sw_cols = _cols(col, NF * len(reg), len(reg), 1)
            for ch in self.ch_names:
                for fn in names:
                    self.keys += [f"{ch}_Sharpwave_{e.title()}_{f}_{fn}" for f, e in reg]
            col += C_ * NF * len(reg)
            if "num_peaks" in used:
                d.sw_numpeaks_cols = _cols(col, NF, 1, 0)
                for ch in self.ch_names:
                    self.keys += [f"{ch}_Sharpwave_num_peaks_{fn}" for fn in names]
                col += C_ * NF
        else:
            pols = ([("Peak")] if d.sw_estimate_peaks else []) + (["Trough"] if d.sw_estimate_troughs else [])
            slots, seen_np = [], False
            for f, e in combos:
                if f == "num_peaks":
                    if seen_np:
                        continue
                    seen_np = True
                    slots.append("{ch}_Sharpwave_num_peaks_{fn}")
                else:
                    slots.append("{ch}_Sharpwave_" + e.title() + "_" + f + "_{fn}")
            npol = len(pols)
            d.sw_cols = _cols(col, NF * len(slots) * npol, len(slots) * npol, npol)
            for ch in self.ch_names:
                for fn in names:
                    for s in slots:
                        self.keys += [s.format(ch=ch, fn=fn) + "_analyze_" + p for p in pols]
            col += C_ * NF * len(slots) * npol
        return col

    # ------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_plan", None) is not None and self._plan.value:
            self.lib.lib.nmx_plan_destroy(self._plan)
            self._plan = C.c_void_p()
        if getattr(self, "_pinned", None) is not None:
            _release_staging(self._pinned)
        self._pinned = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- offset split (include/nmx.h, nmx_engine_dc.inc) -----------------------------------------------------------------
    @property
    def carries_offsets(self) -> bool:
        """The plan can take a recording as (float32 residual, one float64 constant per row)."""
        yes = C.c_int(0)
        self.lib.check(self.lib.lib.nmx_plan_carries_offsets(self._plan, C.byref(yes)))
        return bool(yes.value)

    def set_offsets(self, d) -> None:
        """``d[C_in]`` float64 (None: none): every recording handed over afterwards is split as x = u + d on the host --
        in float64, before the cast to float32.  ``process_batch`` / ``process_window`` choose the constants themselves
        from the first float64 data they see (`_host_offsets`); this is for callers who know better."""
        if d is None:
            self.lib.check(self.lib.lib.nmx_plan_set_offsets(self._plan, None))
            self._dc = np.zeros(self.C_in)
        else:
            d = np.ascontiguousarray(d, dtype=np.float64)
            if d.shape != (self.C_in,):
                raise ValueError(f"expected {self.C_in} offsets, got shape {d.shape}")
            self.lib.check(self.lib.lib.nmx_plan_set_offsets(self._plan, d.ctypes.data))
            self._dc = d.copy()
        self._dc_any = bool(np.any(self._dc != 0.0))

    def offsets(self):
        """(d_in[C_in], d_pre[C]): constants of the input rows (host + learned on the device) and of the pre-processed
        windows the features read."""
        d_in, d_pre = np.zeros(self.C_in), np.zeros(self.C)
        self.lib.check(self.lib.lib.nmx_plan_get_offsets(self._plan, d_in.ctypes.data, d_pre.ctypes.data, None))
        return d_in, d_pre

    def _host_offsets(self, data: np.ndarray):
        """The host constants in force (None: none).  Decided ONCE per stream, by the first data the engine sees -- a
        result must not depend on how the hops were batched: float64 data that needs it (some row's level beyond 64 times
        its spread: below that a float32 cast costs the features nothing) is split at the mean of each row's first
        samples; float32 data is taken as it is (in front of a re-reference the library splits it on the device)."""
        if self._dc is None:
            d = None
            if data.dtype == np.float64 and self.carries_offsets and True:
                # (C order: NumPy's sums follow the memory layout, the constants must not.  The copy lives in a scratch array
                # of this thread and the deviations are formed in place: every half-megabyte temporary is an mmap / munmap pair,
                # and on a host with hundreds of cores and a process with dozens of threads those were 2 ms of a fresh stream)
                w = min(data.shape[1], 256)
                seg = _offset_scratch(data.shape[0], w)
                np.copyto(seg, data[:, :w])
                ok = np.isfinite(seg)
                if ok.all():   # (the same sums over the same C-ordered values as below: the same constants)
                    m = seg.sum(1) / w
                    seg -= m[:, None]
                    np.multiply(seg, seg, out=seg)
                    sd = np.sqrt(seg.sum(1) / w)
                    if np.any(np.abs(m) > 64.0 * sd):
                        d = m
                else:
                    cnt = np.maximum(ok.sum(1), 1)
                    m = np.where(ok, seg, 0.0).sum(1) / cnt
                    sd = np.sqrt(np.where(ok, (seg - m[:, None]) ** 2, 0.0).sum(1) / cnt)
                    if np.any(np.abs(m) > 64.0 * sd):
                        d = np.where(ok.any(1), m, 0.0)
            if d is not None:
                self.set_offsets(d)
            else:
                self._dc = np.zeros(self.C_in)
                self._dc_any = False
        return self._dc if self._dc_any else None

    def reset_state(self) -> None:
        self.lib.check(self.lib.lib.nmx_state_reset(self._plan))
        if self._dc is not None and self._dc_any:
            self.lib.check(self.lib.lib.nmx_plan_set_offsets(self._plan, None))
        self._dc = None
        self._dc_any = False

    def export_state(self) -> bytes:
        n = C.c_int64()
        self.lib.check(self.lib.lib.nmx_state_size(self._plan, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self.lib.check(self.lib.lib.nmx_state_export(self._plan, buf, n.value))
        return buf.raw

    def import_state(self, blob: bytes) -> None:
        self.lib.check(self.lib.lib.nmx_state_import(self._plan, blob, len(blob)))
        # the blob carries the offsets of the stream it came from: adopt them
        st = C.c_int(0)
        d_in = np.zeros(self.C_in)
        self.lib.check(self.lib.lib.nmx_plan_get_offsets(self._plan, d_in.ctypes.data, None, C.byref(st)))
        if st.value & 1:
            host = d_in   # (host offsets set: nothing is learned next to them)
            self._dc, self._dc_any = host, bool(np.any(host != 0.0))
        else:
            self._dc, self._dc_any = np.zeros(self.C_in), False

    def process_window(self, data: np.ndarray, want_nan_mask: bool = False):
        """data[C_in, W] (float64, may be a non-contiguous view) -> float32[n_outputs]."""
        data = np.asarray(data, dtype=np.float64)
        if data.ndim != 2 or data.shape[0] != self.C_in or data.shape[1] != self.W_in:
            raise ValueError(f"expected data of shape ({self.C_in}, {self.W_in}), got {data.shape}")
        if data.strides[1] != 8:
            data = np.ascontiguousarray(data)
        self._host_offsets(data)   # (the library subtracts them from the float64 window itself)
        out = np.empty(self.n_outputs, np.float32)
        mask = np.zeros(self.C_in, np.uint8) if want_nan_mask else None
        self.lib.check(self.lib.lib.nmx_process_window(
            self._plan, data.ctypes.data, _ld(data), out.ctypes.data,
            mask.ctypes.data if mask is not None else None))
        return (out, mask.astype(bool)) if want_nan_mask else out

    def attach_normalizer(self, norm) -> None:
        """Run ``norm`` (a DeviceFeatureNormalizer or None) inside this plan's launch sequence: every batch /
        window comes back already normalised, without a second host round trip (nmx_plan_attach_norm)."""
        self.lib.check(self.lib.lib.nmx_plan_attach_norm(self._plan, norm._h if norm is not None else None))
        self._norm = norm   # keep it alive

    def pinned_empty(self, shape, dtype=np.float32) -> np.ndarray:
        """A page-locked array of the caller's own: inputs / ``out=`` buffers allocated here move at the full PCIe
        rate, asynchronously.  The allocation belongs to the returned array (and its views): it is released
        (nmx_host_free) when the last of them is garbage-collected, never handed to anybody else."""
        import weakref

        shape = tuple(int(x) for x in shape)
        count = int(np.prod(shape))
        n = max(count * np.dtype(dtype).itemsize, 1)
        p = C.c_void_p()
        lib = self.lib
        lib.check(lib.lib.nmx_host_alloc(n, C.byref(p)))
        buf = (C.c_char * n).from_address(p.value)
        weakref.finalize(buf, lib.lib.nmx_host_free, p.value)   # numpy keeps `buf` alive as the base of every view
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

    @property
    def preprocessing_is_identity(self) -> bool:
        """No stage between the incoming rows and the features (no re-reference / channel pick, FIR stage,
        resampler, raw normaliser): the features see nan_to_num(window) itself."""
        d = self.desc
        return not (bool(d.ref_matrix) or d.n_notch_taps or d.n_pre_filters or d.raw_norm_method
                    or self.resample_ratio)

    def process_batch(self, data: np.ndarray, starts: np.ndarray, want_nan_mask: bool = False,
                      staged_output: bool = False, out: np.ndarray | None = None, tap: bool = False):
        """data[C_in, T] host array, starts[n] window start samples -> float32[n, n_outputs].

        ``tap=True`` appends float32[n, C, W] to the result: the pre-processed windows the features were computed
        from (nmx_process_batch_tap) -- the ``data`` argument of ``NMFeature.calc_feature`` for user-registered
        host features (features/feature_processor.py:52-53,80-82).

        A recording that is not contiguous float32 is cast into a page-locked staging array (first axis
        split over a few threads): the host -> device copies then run at the PCIe rate next to the kernels;
        contiguous float32 input is handed over as it is (allocate it with ``pinned_empty`` for the same
        effect).  ``out``: caller's float32[n, n_outputs] buffer.  ``staged_output=True`` returns a VIEW of
        the library's page-locked output staging array (valid until the next call) instead of a fresh
        array -- for callers that convert / consume the rows right away."""
        data = np.asarray(data)
        if data.ndim != 2 or data.shape[0] != self.C_in:
            raise ValueError(f"expected data with {self.C_in} rows, got {data.shape}")
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        n = len(starts)
        dc = self._host_offsets(data)
        if dc is None and _rows_ok(data, np.float32) and _ld(data) >= data.shape[1]:
            x = data   # (rows of a C-ordered float32 array or a view with longer rows: handed over as it is)
        elif data.size >= (1 << 18):
            x = self._pinned.array("x", data.shape, np.float32)
            parallel_cast(x, data, dc, self.lib)
        elif dc is not None:
            # any layout comes in (a transposed (samples, channels) array, a DataFrame's ``to_numpy().T``: the reference
            # takes them all, stream/stream.py:108); the library reads C-ordered float32 rows
            x = np.ascontiguousarray(np.asarray(data, dtype=np.float64) - dc[:, None], dtype=np.float32)
        else:
            x = np.ascontiguousarray(data, dtype=np.float32)
        if out is not None:
            if out.dtype != np.float32 or out.shape != (n, self.n_outputs) or not out.flags.c_contiguous:
                raise ValueError(f"out must be a C-contiguous float32 array of shape ({n}, {self.n_outputs})")
        elif staged_output and n * self.n_outputs >= (1 << 18):
            out = self._pinned.array("out", (n, self.n_outputs), np.float32)
        else:
            out = np.empty((n, self.n_outputs), np.float32)
        if want_nan_mask and data.size >= (1 << 18):   # (page-locked next to page-locked samples / rows: see run_pipelined)
            mask = self._pinned.array("mask", (n, self.C_in), np.uint8)
            mask[:] = 0
        else:
            mask = np.zeros((n, self.C_in), np.uint8) if want_nan_mask else None
        if tap:
            pre = np.empty((n, self.C, self.W), np.float32)
            self.lib.check(self.lib.lib.nmx_process_batch_tap(
                self._plan, x.ctypes.data, _ld(x), x.shape[1], starts.ctypes.data, n,
                out.ctypes.data, mask.ctypes.data if mask is not None else None, 0, None, pre.ctypes.data))
            d_pre = self.offsets()[1]
            if np.any(d_pre != 0.0):   # the windows in float64 with their constants back (NMFeature.calc_feature's argument)
                pre = pre.astype(np.float64) + d_pre[None, :, None]
            return (out, mask.astype(bool), pre) if want_nan_mask else (out, pre)
        self.lib.check(self.lib.lib.nmx_process_batch(
            self._plan, x.ctypes.data, _ld(x), x.shape[1], starts.ctypes.data, n,
            out.ctypes.data, mask.ctypes.data if mask is not None else None, 0, None))
        return (out, mask.astype(bool)) if want_nan_mask else out

    def process_batch_f64(self, data: np.ndarray, starts: np.ndarray, want_nan_mask: bool = False, spare_cols: int = 0):
        """``process_batch`` returning the float64 table the reference's consumers expect, with BOTH conversions next to
        the device work instead of around it (nmx_plan_set_pipeline): a thread converts the recording to float32 slice
        by slice into the page-locked staging array and publishes how far it got -- the library enqueues a chunk's copy
        as soon as the samples it reads are there --, another widens the feature rows to float64 as the chunks land
        (their first touch of a fresh 75 MB table costs more than the arithmetic: it, too, hides under the kernels).
        (Measured round 4, 256 ch x 120 s: conversion 1.9 ms + batch 11.7 ms + widening and first touch 4 - 8 ms in a
        row.)  ``spare_cols``: the table gets that many extra columns behind the features for the caller to fill (the
        reference's frame carries "time" and the target channels behind them, stream/stream.py:319-343 -- written into the
        table they need no column insert)."""
        data = np.asarray(data)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        n, F = len(starts), self.n_outputs
        if data.ndim != 2 or data.shape[0] != self.C_in:
            raise ValueError(f"expected data with {self.C_in} rows, got {data.shape}")
        small = data.size < (1 << 20) or n < 64 or os.environ.get("NMX_PIPELINE", "1") == "0"
        if small or (_rows_ok(data, np.float32) and _ld(data) >= data.shape[1] and self._host_offsets(data) is None):
            res = self.process_batch(data, starts, want_nan_mask=want_nan_mask, staged_output=True)
            out = res[0] if want_nan_mask else res
            o64 = table_empty((n, F + spare_cols))
            parallel_cast(o64[:, :F], out, None, self.lib)
            return (o64, res[1]) if want_nan_mask else o64
        dc = self._host_offsets(data)
        x = self._pinned.array("x", data.shape, np.float32)
        o64 = table_empty((n, F + spare_cols))
        mask = self.run_pipelined(x, starts, o64, stage=lambda a, b: parallel_cast(x[:, a:b], data[:, a:b], dc, self.lib),
                                  want_nan_mask=want_nan_mask)
        return (o64, mask) if want_nan_mask else o64

    def pipeline_counters(self) -> np.ndarray:
        """The int64 counters of ``nmx_plan_set_pipeline`` in page-locked memory, zeroed: [0] samples of the staging array
        in place (written by whoever stages), [8] feature rows landed (written by the library) -- own cache lines."""
        ctr = self._pinned.array("ctr", (16,), np.int64)
        ctr[:] = 0
        return ctr

    def pipeline_edges(self, starts: np.ndarray, T: int) -> list[int]:
        """Sample counts at which to publish progress while staging a recording: what the library's chunks read
        (nmx_engine_run.inc: a short first chunk -- 128 hops, or the fill phase of the burst history, ~320 --, then 512
        hops at a time)."""
        n = len(starts)
        hops = [h for h in (128, 320) if h < n] + list(range(320 + 512, n, 512)) + [n]
        # (finer slices were measured and lose: publishing the first chunk in 8 k-sample pieces, copied as they arrive,
        # costs more in staging calls than the earlier start buys -- Stream.run 17.2 -> 17.9 ms, two plans 12.5 -> 13.6)
        return sorted(set([0] + [int(min(T, starts[h - 1] + self.W_in)) for h in hops] + [T]))

    def run_pipelined(self, x: np.ndarray, starts: np.ndarray, table: np.ndarray, runs: np.ndarray | None = None,
                      stage=None, ctr: np.ndarray | None = None, want_nan_mask: bool = False):
        """One batch with staging and widening NEXT to the device work.  ``x`` float32 [C_in, T]: the (page-locked)
        staging array the library copies from; it is filled either by ``stage(a, b)`` -- called here, on a thread of
        this call, for consecutive sample ranges -- or by the caller, who then owns ``ctr`` (``pipeline_counters()``) and
        publishes in ``ctr[0]`` how many samples of EVERY row are in place (a multi-device stream stages all its parts
        from one thread).  The rows are widened as the chunks land into the float64 ``table``: columns
        ``runs[k] = (first column in table, first column of the engine's row, length)``, None: the whole row from column
        0.  -> the NaN mask (bool [n, C_in]) or None."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        n, T = len(starts), x.shape[1]
        if not (_rows_ok(x, np.float32) and x.shape[0] == self.C_in and _rows_ok(table, np.float64) and table.shape[0] == n):
            raise ValueError("run_pipelined: float32 staging rows and a float64 table with one row per window")
        if runs is None:
            runs = np.array([[0, 0, self.n_outputs]], dtype=np.int64)
        runs = np.ascontiguousarray(runs, dtype=np.int64).reshape(-1, 3)
        out = self._pinned.array("out", (n, self.n_outputs), np.float32)
        # (page-locked like `out`: a copy into PAGEABLE memory is synchronous for the calling thread -- the library's host
        # loop stood at the mask of chunk k - 1 until that chunk was complete, and the samples of chunk k + 1 waited with it)
        mask = self._pinned.array("mask", (n, self.C_in), np.uint8) if want_nan_mask else None
        if mask is not None:
            mask[:] = 0
        own = ctr is None
        if own:
            ctr = self.pipeline_counters()
        failed: list = []
        lib = self.lib

        def convert():
            try:
                edges = self.pipeline_edges(starts, T)
                for a, b in zip(edges[:-1], edges[1:]):
                    stage(a, b)
                    ctr[0] = b
            except BaseException as e:   # noqa: BLE001 -- reported by the caller's thread
                failed.append(e)
                ctr[0] = T

        def widen():
            done = 0
            try:
                while done < n and not failed:
                    d = int(ctr[8])
                    if d > done:
                        lib.check(lib.lib.nmx_host_widen_rows(table.ctypes.data, table.strides[0] // 8, out.ctypes.data,
                                                              self.n_outputs, done, d, runs.ctypes.data, len(runs), 0))
                        done = d
                    else:
                        time.sleep(0.0001)
            except BaseException as e:   # noqa: BLE001
                failed.append(e)

        import threading

        lib.check(lib.lib.nmx_plan_set_pipeline(self._plan, ctr.ctypes.data, ctr.ctypes.data + 64))
        jobs = [threading.Thread(target=widen, daemon=True)]
        if stage is not None:
            jobs.append(threading.Thread(target=convert, daemon=True))
        elif own:
            ctr[0] = T   # (nobody stages: the array is complete)
        for j in jobs:
            j.start()
        try:
            lib.check(lib.lib.nmx_process_batch(
                self._plan, x.ctypes.data, _ld(x), T, starts.ctypes.data, n,
                out.ctypes.data, mask.ctypes.data if mask is not None else None, 0, None))
        except BaseException:
            if own:
                ctr[0] = T   # (let the threads run out)
            ctr[8] = n
            failed.append(None)
            raise
        finally:
            lib.lib.nmx_plan_set_pipeline(self._plan, None, None)
            for j in jobs:
                j.join()
        if failed:
            raise failed[0]
        return mask.astype(bool) if want_nan_mask else None

    def process_batch_device(self, x_ptr: int, ldx: int, n_samples: int, starts: np.ndarray,
                             out_ptr: int, mask_ptr: int | None = None, stream: int | None = None) -> None:
        """Device-resident variant: raw pointers on the plan's device (e.g. torch ``data_ptr()``)."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        self.lib.check(self.lib.lib.nmx_process_batch(
            self._plan, x_ptr, ldx, n_samples, starts.ctypes.data, len(starts), out_ptr, mask_ptr,
            1, stream))

    def preprocess_window(self, data: np.ndarray) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.float64)
        if data.ndim != 2 or data.shape != (self.C_in, self.W_in):
            raise ValueError(f"expected data of shape ({self.C_in}, {self.W_in}), got {data.shape}")
        y = np.empty((self.C, self.W), np.float64)
        self.lib.check(self.lib.lib.nmx_preprocess_window(self._plan, data.ctypes.data, data.shape[1],
                                                         y.ctypes.data, self.W))
        return y

    def filter_window(self, data: np.ndarray) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.float64)
        y = np.empty((self.C, int(self.desc.n_filters), self.W), np.float64)
        self.lib.check(self.lib.lib.nmx_filter_window(self._plan, data.ctypes.data, data.shape[1],
                                                     y.ctypes.data))
        return y

    def timing_ms(self, which: int = 0) -> float:
        ms = C.c_float()
        self.lib.check(self.lib.lib.nmx_last_timing_ms(self._plan, which, C.byref(ms)))
        return float(ms.value)

    def kernels(self, which: int) -> str:
        """Kernels the last batch launched in stage ``which`` (1 prep, 2 time/osc, 3 FIR bank, 4 bursts,
        5 sharp waves, 6 the FIR-bank filters left to a second launch: taps too long for the M = 1536 kernel),
        named as rocprofv3 prints them."""
        buf = C.create_string_buffer(512)
        self.lib.check(self.lib.lib.nmx_last_kernels(self._plan, which, buf, 512))
        return buf.value.decode()

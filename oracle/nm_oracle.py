"""CPU oracle: float64 NumPy/SciPy restatement of py_neuromodulation's per-hop hot path.

TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/__init__.py).  Every class mirrors the
reference's plugin signature ``cls(settings, ch_names, sfreq).calc_feature(data) -> dict``
so the parity tests read like the reference's own tests.  ``settings`` is duck-typed: the
reference's ``NMSettings`` (build container only) and ``py_neuromodulation_amd.settings``
both work.  Citations are into /root/reference/py_neuromodulation/.

The restatement is written from the arithmetic in SURVEY.md Appendix A, not copied: it is
vectorised over channels/bands where the reference loops in Python, computes the FIR bank
with ONE forward FFT per channel (the reference tiles data and taps, filter/mne_filter.py:
110-116) and replaces scipy.ndimage.label bookkeeping in Bursts by run-length arithmetic.
Results agree with the reference to float64 rounding (tests/test_oracle_golden.py).
"""

from __future__ import annotations

import math
from collections.abc import Sequence

import numpy as np
from scipy import fft as sp_fft

from . import mne_restated

# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------


def _enabled(selector) -> list[str]:
    """BoolSelector.get_enabled() (utils/types.py:134-140) for either settings flavour."""
    return list(selector.get_enabled())


def _band_items(settings):
    return [(name, (fr[0], fr[1])) for name, fr in settings.frequency_ranges_hz.items()]


def window_schedule(n_samples: int, sfreq: float, feat_hz: float, seg_ms: float):
    """stream/generator.py:34-53 + stream/stream.py:298,310.

    Returns (start_idx[int64], end_idx[int64], time_ms[float64]) for every window the
    generator yields.  Float stride/segment, ``int()`` truncation, stop when end > T.
    time_ms = ceil(timestamps[-1] * 1000 + 1), timestamps = arange(start, end) / sfreq.
    """
    seg = seg_ms / 1000 * sfreq
    stride = sfreq / feat_hz
    starts, ends, times = [], [], []
    k = 0
    while True:
        start = stride * k
        end = start + seg
        k += 1
        if int(end) > n_samples:
            break
        ts = np.arange(start, end) / sfreq
        starts.append(int(start))
        ends.append(int(end))
        times.append(math.ceil(ts[-1] * 1000 + 1))
    return (np.asarray(starts, np.int64), np.asarray(ends, np.int64),
            np.asarray(times, np.float64))


def fir_bank_apply(data: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """filter/mne_filter.py:110-126: y[c,b,n] = sum_k h[b,k] x[c, n + (L-1)//2 - k].

    data (C, W), taps (B, L) -> (C, B, W); zero outside [0, W) ("same" centred on the
    full convolution, also when L > W).
    """
    data = np.atleast_2d(np.asarray(data, np.float64))
    taps = np.atleast_2d(np.asarray(taps, np.float64))
    W, L = data.shape[-1], taps.shape[-1]
    n = sp_fft.next_fast_len(W + L - 1, real=True)
    X = sp_fft.rfft(data, n=n, axis=-1)
    H = sp_fft.rfft(taps, n=n, axis=-1)
    full = sp_fft.irfft(X[:, None, :] * H[None, :, :], n=n, axis=-1)
    s = (L - 1) // 2
    return full[:, :, s : s + W]


def _var(x, axis=-1):
    return np.var(x, axis=axis)


def hjorth_params(x: np.ndarray):
    """features/hjorth_raw.py:24-34 (also bandpower.py:185-207) along the last axis."""
    d1 = np.diff(x, axis=-1)
    d2 = np.diff(d1, axis=-1)
    v0, v1, v2 = _var(x), _var(d1), _var(d2)
    with np.errstate(divide="ignore", invalid="ignore"):
        mobility = np.sqrt(v1 / v0)
        complexity = np.sqrt(v2 / v1) / mobility
    return v0, mobility, complexity


_NAN_EST = {"mean": np.nanmean, "median": np.nanmedian, "std": np.nanstd, "max": np.nanmax}


def find_peaks_distance(x: np.ndarray, distance: float) -> np.ndarray:
    """scipy.signal.find_peaks(x, distance=distance)[0], restated.

    Strict local maxima, plateaus -> midpoint (l + r) // 2, end points never peaks;
    then visit peaks by decreasing height (stable: later index wins ties, as SciPy's
    argsort order traversed from the end) and drop neighbours closer than ceil(distance).
    """
    n = len(x)
    peaks = []
    i = 1
    i_max = n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            ahead = i + 1
            while ahead < i_max and x[ahead] == x[i]:
                ahead += 1
            if x[ahead] < x[i]:
                left, right = i, ahead - 1
                peaks.append((left + right) // 2)
                i = ahead
        i += 1
    peaks = np.asarray(peaks, dtype=np.intp)
    if peaks.size == 0 or distance is None:
        return peaks
    dist = math.ceil(distance)
    keep = np.ones(peaks.size, dtype=bool)
    order = np.argsort(x[peaks])  # SciPy uses the default (unstable) argsort: ties within
    # `distance` resolve implementation-dependently there too
    for i in range(peaks.size - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < dist:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < peaks.size and peaks[k] - peaks[j] < dist:
            keep[k] = False
            k += 1
    return peaks[keep]


# --------------------------------------------------------------------------------------
# time-domain features
# --------------------------------------------------------------------------------------


class Hjorth:
    """features/hjorth_raw.py:18-42."""

    def __init__(self, settings, ch_names: Sequence[str], sfreq: float) -> None:
        self.ch_names = list(ch_names)

    def calc_feature(self, data: np.ndarray) -> dict:
        data = np.asarray(data, np.float64)
        d1 = np.diff(data, axis=-1)
        v0, v1, v2 = _var(data), _var(d1), _var(np.diff(d1, axis=-1))
        with np.errstate(divide="ignore", invalid="ignore"):
            act = np.nan_to_num(v0)
            mob = np.nan_to_num(np.sqrt(v1 / v0))
            # NB the reference divides by the nan_to_num'ed mobility (hjorth_raw.py:31-34)
            comp = np.nan_to_num(np.sqrt(v2 / v1) / mob)
        out = {}
        for i, ch in enumerate(self.ch_names):
            out[f"{ch}_RawHjorth_Activity"] = act[i]
            out[f"{ch}_RawHjorth_Mobility"] = mob[i]
            out[f"{ch}_RawHjorth_Complexity"] = comp[i]
        return out


class Raw:
    """features/hjorth_raw.py:45-57."""

    def __init__(self, settings, ch_names, sfreq) -> None:
        self.ch_names = list(ch_names)

    def calc_feature(self, data):
        return {f"{ch}_raw": data[i, -1] for i, ch in enumerate(self.ch_names)}


class LineLength:
    """features/linelength.py:11-21: mean(|diff| / (W - 1)) = sum|dx| / (W - 1)^2."""

    def __init__(self, settings, ch_names, sfreq) -> None:
        self.ch_names = list(ch_names)

    def calc_feature(self, data):
        W = data.shape[1]
        ll = np.mean(np.abs(np.diff(data, axis=-1)) / (W - 1), axis=-1)
        return {f"{ch}_LineLength": ll[i] for i, ch in enumerate(self.ch_names)}


# --------------------------------------------------------------------------------------
# oscillatory features (features/oscillatory.py)
# --------------------------------------------------------------------------------------


class _Oscillatory:
    name = ""

    def _common(self, settings, ch_names, sfreq, osc_settings):
        self.s = osc_settings
        self.sfreq = int(sfreq)
        self.ch_names = list(ch_names)
        assert self.s.windowlength_ms <= settings.segment_length_features_ms
        self.estimators = _enabled(self.s.features)

    def _emit(self, Z, Zband_fn, freqs_for_psd, psd_fn):
        out = {}
        with np.errstate(all="ignore"):
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for band, idx in self.idx_range:
                    Zb = Zband_fn(idx)
                    for est in self.estimators:
                        res = _NAN_EST[est](Zb, axis=self._est_axis)
                        for i, ch in enumerate(self.ch_names):
                            out[f"{ch}_{self.name}_{band}_{est}"] = res[i]
        if self.s.return_spectrum:
            for i, ch in enumerate(self.ch_names):
                for k, f in enumerate(freqs_for_psd):
                    out[f"{ch}_{self.name}_psd_{int(f)}"] = psd_fn(i, k)
        return out


class FFT(_Oscillatory):
    """features/oscillatory.py:58-119: |rfft(x[:, -N:])| -> log10 -> band estimators, bins [lo, hi)."""

    name = "fft"
    _est_axis = 1

    def __init__(self, settings, ch_names, sfreq) -> None:
        self._common(settings, ch_names, sfreq, settings.fft_settings)
        self.N = int(np.floor(self.s.windowlength_ms / 1000 * sfreq))
        self.freqs = sp_fft.rfftfreq(self.N, 1 / np.floor(self.sfreq))
        self.idx_range = [
            (b, np.where((self.freqs >= lo) & (self.freqs < hi))[0])
            for b, (lo, hi) in _band_items(settings)
        ]

    def spectrum(self, data):
        Z = np.abs(sp_fft.rfft(np.asarray(data, np.float64)[:, -self.N:], axis=-1))
        if self.s.log_transform:
            with np.errstate(divide="ignore"):
                Z = np.log10(Z)
        return Z

    def calc_feature(self, data):
        Z = self.spectrum(data)
        return self._emit(Z, lambda idx: Z[:, idx], self.freqs, lambda i, k: Z[i][k])


def welch_psd(data: np.ndarray, fs: int, nperseg: int) -> np.ndarray:
    """scipy.signal.welch(x, fs, "hann", nperseg, noverlap=None) restated.

    hann (periodic), 50 % overlap, per-segment mean removal, density scaling
    1 / (fs * sum(w^2)), one-sided doubling, mean over segments.
    """
    x = np.asarray(data, np.float64)
    W = x.shape[-1]
    nperseg = min(nperseg, W)
    step = nperseg - nperseg // 2
    n = np.arange(nperseg)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / nperseg)
    scale = 1.0 / (fs * np.sum(w * w))
    starts = range(0, W - nperseg + 1, step)
    acc = 0.0
    for s in starts:
        seg = x[..., s : s + nperseg]
        seg = seg - seg.mean(axis=-1, keepdims=True)
        P = np.abs(sp_fft.rfft(seg * w, axis=-1)) ** 2 * scale
        acc = acc + P
    P = acc / len(starts)
    if nperseg % 2 == 0:
        P[..., 1:-1] *= 2
    else:
        P[..., 1:] *= 2
    return P


class Welch(_Oscillatory):
    """features/oscillatory.py:122-182 (nperseg = sfreq samples, bins [lo, hi))."""

    name = "welch"
    _est_axis = 1

    def __init__(self, settings, ch_names, sfreq) -> None:
        self._common(settings, ch_names, sfreq, settings.welch_settings)
        self.freqs = sp_fft.rfftfreq(self.sfreq, 1 / self.sfreq)
        self.idx_range = [
            (b, np.where((self.freqs >= lo) & (self.freqs < hi))[0])
            for b, (lo, hi) in _band_items(settings)
        ]

    def spectrum(self, data):
        Z = welch_psd(data, self.sfreq, self.sfreq)
        if self.s.log_transform:
            with np.errstate(divide="ignore"):
                Z = np.log10(Z)
        return Z

    def calc_feature(self, data):
        Z = self.spectrum(data)
        return self._emit(Z, lambda idx: Z[:, idx], self.freqs, lambda i, k: Z[i][k])


def stft_mag(data: np.ndarray, nperseg: int) -> np.ndarray:
    """|scipy.signal.stft(x, window="hamming", nperseg, boundary="even")| restated.

    noverlap = nperseg // 2, nfft = nperseg, padded=True, scaling="spectrum" (/ sum(w)),
    even extension by nperseg // 2 on each side.  Returns (C, nperseg // 2 + 1, n_seg).
    """
    x = np.asarray(data, np.float64)
    h = nperseg // 2
    step = nperseg - h
    left = x[..., h:0:-1]
    right = x[..., -2 : -h - 2 : -1]
    xe = np.concatenate([left, x, right], axis=-1)
    nadd = (-(xe.shape[-1] - nperseg) % step) % nperseg
    if nadd:
        xe = np.concatenate([xe, np.zeros(xe.shape[:-1] + (nadd,))], axis=-1)
    n = np.arange(nperseg)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * n / nperseg)  # periodic hamming
    nseg = (xe.shape[-1] - nperseg) // step + 1
    out = np.empty(x.shape[:-1] + (nperseg // 2 + 1, nseg))
    for m in range(nseg):
        seg = xe[..., m * step : m * step + nperseg] * w
        out[..., m] = np.abs(sp_fft.rfft(seg, axis=-1)) / w.sum()
    return out


class STFT(_Oscillatory):
    """features/oscillatory.py:185-250 (nperseg = windowlength_ms *samples*, bins [lo, hi])."""

    name = "stft"
    _est_axis = (1, 2)

    def __init__(self, settings, ch_names, sfreq) -> None:
        self._common(settings, ch_names, sfreq, settings.stft_settings)
        self.nperseg = int(self.s.windowlength_ms)
        self.freqs = sp_fft.rfftfreq(self.nperseg, 1 / self.sfreq)
        self.idx_range = [
            (b, np.where((self.freqs >= lo) & (self.freqs <= hi))[0])
            for b, (lo, hi) in _band_items(settings)
        ]

    def spectrum(self, data):
        # scipy.signal.stft shrinks nperseg to the input length (with a warning); the band indices keep the
        # nominal grid (oscillatory.py:201-210), so they select other frequencies or raise IndexError
        Z = stft_mag(data, min(self.nperseg, np.shape(data)[-1]))
        if self.s.log_transform:
            with np.errstate(divide="ignore"):
                Z = np.log10(Z)
        return Z

    def calc_feature(self, data):
        Z = self.spectrum(data)
        return self._emit(Z, lambda idx: Z[:, idx, :], self.freqs,
                          lambda i, k: Z[i].mean(axis=1)[k])


# --------------------------------------------------------------------------------------
# FIR bank + BandPower (filter/mne_filter.py, features/bandpower.py)
# --------------------------------------------------------------------------------------


def design_bank(f_ranges, sfreq, filter_length=None, l_trans=4, h_trans=4) -> np.ndarray:
    """filter/mne_filter.py:35-80: create_filter per band, auto-length fallback on ValueError."""
    if filter_length is None:
        filter_length = sfreq - 1
    if isinstance(filter_length, float):
        filter_length = int(filter_length)
    bank = []
    for lo, hi in f_ranges:
        try:
            h = mne_restated.create_filter(None, sfreq, lo, hi, filter_length=filter_length,
                                           l_trans_bandwidth=l_trans, h_trans_bandwidth=h_trans)
        except ValueError:
            h = mne_restated.create_filter(None, sfreq, lo, hi)
        bank.append(h)
    return np.vstack(bank)


class KalmanWNA:
    """filter/kalman_filter.py:45-78 (white-noise-acceleration model) with the predict / update
    steps of filter/kalman_filter_external.py:466-590 (filterpy, Joseph-form covariance)."""

    def __init__(self, Tp: float, sigma_w: float, sigma_v: float) -> None:
        self.x = np.array([0.0, 1.0])
        self.F = np.array([[1.0, Tp], [0.0, 1.0]])
        self.H = np.array([[1.0, 0.0]])
        self.R = sigma_v
        self.Q = np.array([[sigma_w**2 * Tp**3 / 3, sigma_w**2 * Tp**2 / 2],
                           [sigma_w**2 * Tp**2 / 2, sigma_w**2 * Tp]])
        self.P = np.cov([[1, 0], [0, 1]])

    def step(self, z: float) -> float:
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + self.Q
        y = np.atleast_1d(z) - self.H @ self.x
        PHT = self.P @ self.H.T
        S = self.H @ PHT + self.R
        K = PHT @ np.linalg.inv(S)
        self.x = self.x + K @ y
        I_KH = np.eye(2) - K @ self.H
        self.P = I_KH @ self.P @ I_KH.T + (K * self.R) @ K.T
        return self.x[0]


class BandPower:
    """features/bandpower.py:98-207 incl. the optional Kalman smoothing of the activity (:147-163)."""

    def __init__(self, settings, ch_names, sfreq, taps: np.ndarray | None = None) -> None:
        self.s = settings.bandpass_filter_settings
        self.sfreq = sfreq
        self.ch_names = list(ch_names)
        bands = _band_items(settings)
        self.band_names = [b for b, _ in bands]
        self.taps = design_bank([r for _, r in bands], sfreq) if taps is None else taps
        self.seglens = [
            int(np.floor(sfreq / 1000 * self.s.segment_lengths_ms[b])) for b in self.band_names
        ]
        self.feats = _enabled(self.s.bandpower_features)
        self.kf = {}
        if getattr(self.s, "kalman_filter", False):
            ks = settings.kalman_filter_settings
            for band in ks.frequency_bands:
                for ch in self.ch_names:
                    self.kf[f"{ch}_bandpass_activity_{band}"] = KalmanWNA(ks.Tp, ks.sigma_w, ks.sigma_v)

    def calc_feature(self, data):
        y = fir_bank_apply(data, self.taps)  # (C, B, W)
        out = {}
        vals = {}
        with np.errstate(divide="ignore", invalid="ignore"):
            for bi, seglen in enumerate(self.seglens):
                t = y[:, bi, -seglen:]
                act, mob, comp = hjorth_params(t)
                if self.s.log_transform:
                    act = np.log10(act)
                vals[bi] = {"activity": act, "mobility": mob, "complexity": comp}
        for ci, ch in enumerate(self.ch_names):
            for bi, band in enumerate(self.band_names):
                for f in self.feats:
                    key = f"{ch}_bandpass_{f}_{band}"
                    v = vals[bi][f][ci]
                    if f == "activity" and key in self.kf:   # before nan_to_num (:188-197)
                        v = self.kf[key].step(v)
                    out[key] = np.nan_to_num(v)
        return out


# --------------------------------------------------------------------------------------
# Bursts (features/bursts.py)
# --------------------------------------------------------------------------------------


def analytic_envelope(y: np.ndarray) -> np.ndarray:
    """|scipy.signal.hilbert(y)| along the last axis (exact length-W analytic signal)."""
    W = y.shape[-1]
    Y = sp_fft.fft(y, axis=-1)
    h = np.zeros(W)
    if W % 2 == 0:
        h[0] = h[W // 2] = 1
        h[1 : W // 2] = 2
    else:
        h[0] = 1
        h[1 : (W + 1) // 2] = 2
    return np.abs(sp_fft.ifft(Y * h, axis=-1))


def burst_stats(env: np.ndarray, thr: np.ndarray, sfreq: float, seg_s: float) -> dict:
    """features/bursts.py:175-258 for env (..., W) and thr (...,): run-length arithmetic.

    Returns dict of arrays shaped like ``thr``.
    """
    shp = thr.shape
    W = env.shape[-1]
    e = env.reshape(-1, W)
    t = thr.reshape(-1)
    n = e.shape[0]
    res = {k: np.zeros(n) for k in ("duration_mean", "duration_max", "amplitude_mean",
                                   "amplitude_max", "burst_rate_per_s", "in_burst")}
    for r in range(n):
        b = e[r] >= t[r]
        d = np.diff(np.concatenate(([False], b)).astype(np.int8))
        starts = np.flatnonzero(d == 1)
        ends = np.flatnonzero(d == -1)  # exclusive end index of each finished run
        n_trans = np.count_nonzero(d)
        num_bursts = n_trans // 2
        if num_bursts:
            res["duration_mean"][r] = b.sum() / num_bursts / sfreq
        # valid runs = runs that do not include the last sample
        nv = len(ends)
        if nv:
            lens = ends - starts[:nv]
            res["duration_max"][r] = lens.max() / sfreq
            means = np.array([e[r][s:en].mean() for s, en in zip(starts[:nv], ends)])
            res["amplitude_mean"][r] = means.mean()
        res["amplitude_max"][r] = (e[r] * b).max()
        res["in_burst"][r] = float(b[-1])
    res["burst_rate_per_s"] = res["duration_mean"] / seg_s
    return {k: v.reshape(shp) for k, v in res.items()}


class Bursts:
    """features/bursts.py:60-298.  Stateful: ring buffer of envelopes per (channel, band)."""

    def __init__(self, settings, ch_names, sfreq, taps: np.ndarray | None = None) -> None:
        self.s = settings.bursts_settings
        for fb in self.s.frequency_bands:
            if fb not in settings.frequency_ranges_hz:
                raise ValueError(f"bursting {fb} needs to be defined in settings['frequency_ranges_hz']")
        self.sfreq = sfreq
        self.ch_names = list(ch_names)
        self.seg_s = settings.segment_length_features_ms / 1000
        self.samples_overlap = int(sfreq * self.seg_s / settings.sampling_rate_features_hz)
        self.band_names = list(self.s.frequency_bands)
        ranges = [(settings.frequency_ranges_hz[b][0], settings.frequency_ranges_hz[b][1])
                  for b in self.band_names]
        self.taps = design_bank(ranges, sfreq) if taps is None else taps
        self.n_ring = int(sfreq * self.s.time_duration_s)
        self.buffer = np.empty((len(self.ch_names), len(self.band_names), 0))
        self.batch = 0
        self.feats = _enabled(self.s.burst_features)

    def envelope(self, data):
        return analytic_envelope(fir_bank_apply(data, self.taps))

    def update_threshold(self, new: np.ndarray) -> np.ndarray:
        """Append ``new`` (C, B, n_new) to the ring and return the percentile threshold.

        REFERENCE QUIRK (features/bursts.py:3-6,156-173): the threshold is computed with
        NumPy's *private* ``_quantile``, which ``partition``s its argument IN PLACE, and the
        argument is ``self.data_buffer`` itself.  From the second call on the "ring" is
        therefore no longer time-ordered: positions [0, lo] hold (in implementation-defined
        order) elements <= the lower interpolation neighbour s[lo], lo = floor(q (n - 1)).
        When the buffer overflows by t, ``[..., -n_ring:]`` drops the first t positions,
        i.e. t elements that are all <= s[lo] -- never the oldest ones, never one of the
        top n - 1 - lo.  WHICH of the small elements go is implementation-defined (NumPy's
        introselect / AVX-512 quickselect), but irrelevant: every later threshold is an
        order statistic at a fixed distance from the TOP of a multiset whose top part is
        never removed, so it only depends on how many elements lie below it.  The
        well-defined equivalent used here: on overflow drop the t smallest elements.
        Consequence (reproduced, not fixed): once the ring is full the threshold is a
        running order statistic over the WHOLE history and never decreases.
        """
        buf = np.concatenate((self.buffer, new), axis=2)
        t = buf.shape[-1] - self.n_ring
        if t > 0:
            buf = np.partition(buf, t - 1, axis=-1)[:, :, t:]
        self.buffer = buf
        return np.quantile(buf, self.s.threshold / 100, axis=-1)

    def calc_feature(self, data):
        env = self.envelope(data)
        n_new = env.shape[-1] if self.batch == 0 else self.samples_overlap
        self.batch += 1
        thr = self.update_threshold(env[:, :, -n_new:])
        st = burst_stats(env, thr, self.sfreq, self.seg_s)
        self.last_thr, self.last_env = thr, env   # (read by the conditioning report of tests/parity.py)
        out = {}
        for ci, ch in enumerate(self.ch_names):
            for bi, fb in enumerate(self.band_names):
                for f in self.feats:
                    if f == "duration":
                        out[f"{ch}_bursts_{fb}_duration_mean"] = st["duration_mean"][ci, bi]
                        out[f"{ch}_bursts_{fb}_duration_max"] = st["duration_max"][ci, bi]
                    elif f == "amplitude":
                        out[f"{ch}_bursts_{fb}_amplitude_mean"] = st["amplitude_mean"][ci, bi]
                        out[f"{ch}_bursts_{fb}_amplitude_max"] = st["amplitude_max"][ci, bi]
                    elif f == "burst_rate_per_s":
                        out[f"{ch}_bursts_{fb}_burst_rate_per_s"] = st["burst_rate_per_s"][ci, bi]
                    elif f == "in_burst":
                        out[f"{ch}_bursts_{fb}_in_burst"] = st["in_burst"][ci, bi]
        return out


# --------------------------------------------------------------------------------------
# Sharp waves (features/sharpwaves.py)
# --------------------------------------------------------------------------------------

_SW_FEATURE_ORDER = ["peak_left", "peak_right", "num_peaks", "trough", "width", "prominence",
                     "interval", "decay_time", "rise_time", "sharpness", "rise_steepness",
                     "decay_steepness", "slope_ratio"]
_SW_EST_ORDER = ["mean", "median", "max", "min", "var"]
_SW_EST = {
    "mean": lambda a: np.add.reduce(np.asarray(a, np.float64)) / np.size(a),
    "median": np.median, "max": np.max, "min": np.min, "var": np.var,
}


def sharpwave_design(settings, sfreq):
    """features/sharpwaves.py:121-146: auto-length band-pass per filter range."""
    names, taps = [], []
    for fr in settings.sharpwave_analysis_settings.filter_ranges_hz:
        assert fr[1] < sfreq
        names.append(f"range_{fr[0]:.0f}_{fr[1]:.0f}")
        taps.append(mne_restated.create_filter(None, sfreq, fr[0], fr[1]))
    return names, taps


def analyze_waveform(z: np.ndarray, sfreq: float, dist_peaks: float, dist_troughs: float,
                     need: set[str]) -> dict:
    """features/sharpwaves.py:330-465 for one 1-D series ``z`` (already sign-flipped)."""
    W = len(z)
    peaks = find_peaks_distance(z, dist_peaks)
    troughs = find_peaks_distance(-z, dist_troughs)
    ptr = first_valid = last_valid = 0
    left, right = [], []
    for i in range(len(troughs)):
        while ptr < peaks.size and peaks[ptr] < troughs[i]:
            ptr += 1
        if ptr - 1 < 0:
            first_valid = i + 1
            continue
        if ptr == peaks.size:
            continue
        last_valid = i
        left.append(peaks[ptr - 1])
        right.append(peaks[ptr])
    troughs = troughs[first_valid : last_valid + 1]
    left = np.asarray(left, dtype=int)
    right = np.asarray(right, dtype=int)
    ms = 1000 / sfreq
    res: dict = {}
    res["peak_left"], res["peak_right"], res["trough"] = z[left], z[right], z[troughs]
    if "interval" in need:
        res["interval"] = np.concatenate((np.zeros(1), np.diff(troughs))) * ms
    if "sharpness" in need:
        s = int(5 * ms)
        tv = troughs[np.logical_and(troughs - s > 0, troughs + s < W)]
        res["sharpness"] = z[tv] - 0.5 * (z[tv - s] + z[tv + s])
    if "num_peaks" in need:
        res["num_peaks"] = [troughs.shape[0]]
    if need & {"rise_steepness", "decay_steepness", "slope_ratio"}:
        st = np.concatenate((np.zeros(1), np.diff(z)))
        n = troughs.shape[0]
        rise = np.zeros(n)
        decay = np.zeros(n)
        for i in range(n):
            rise[i] = np.max(np.abs(st[left[i] : troughs[i] + 1]))
            decay[i] = np.max(np.abs(st[troughs[i] : right[i] + 1]))
        res["rise_steepness"], res["decay_steepness"] = rise, decay
        res["slope_ratio"] = rise - decay
    if "prominence" in need:
        res["prominence"] = np.abs((res["peak_right"] + res["peak_left"]) / 2 - res["trough"])
    if "decay_time" in need:
        res["decay_time"] = (left - troughs) * ms
    if "rise_time" in need:
        res["rise_time"] = (right - troughs) * ms
    if "width" in need:
        res["width"] = right - left
    return res


class SharpwaveAnalyzer:
    """features/sharpwaves.py:100-328."""

    def __init__(self, settings, ch_names, sfreq, taps: list | None = None) -> None:
        self.s = settings.sharpwave_analysis_settings
        self.sfreq = sfreq
        self.ch_names = list(ch_names)
        self.filter_names, designed = (sharpwave_design(settings, sfreq) if taps is None else
                                       ([f"range_{fr[0]:.0f}_{fr[1]:.0f}"
                                         for fr in self.s.filter_ranges_hz], taps))
        self.taps = [np.asarray(t, np.float64) for t in designed]
        self.used = [f for f in _SW_FEATURE_ORDER if getattr(self.s.sharpwave_features, f)]
        est = self.s.estimator
        self.est_of = {f: [e for e in _SW_EST_ORDER if f in getattr(est, e)] for f in self.used}
        for f in self.used:
            assert self.est_of[f], f"Add estimator key for {f}"
        self.combos = [(f, e) for f in self.used for e in self.est_of[f]]

    def filtered(self, data):
        return np.stack([fir_bank_apply(data, t[None])[:, 0] for t in self.taps], axis=1)

    def calc_feature(self, data):
        y = self.filtered(data)
        need = set(self.used)
        per_key: dict[str, dict[str, float]] = {}
        pol = []
        if self.s.detect_peaks.estimate:
            pol.append(("Peak", 1.0))
        if self.s.detect_troughs.estimate:
            pol.append(("Trough", -1.0))
        dp = self.s.detect_troughs.distance_peaks_ms     # sharpwaves.py:339-344 always reads
        dt = self.s.detect_troughs.distance_troughs_ms   # the detect_troughs block
        for ci, ch in enumerate(self.ch_names):
            for fi, fname in enumerate(self.filter_names):
                for pname, sign in pol:
                    r = analyze_waveform(sign * y[ci, fi], self.sfreq, dp, dt, need)
                    for f, e in self.combos:
                        if f == "num_peaks":
                            per_key.setdefault(f"{ch}_Sharpwave_{f}_{fname}", {})[pname] = r[f][0]
                            continue
                        v = r[f]
                        val = _SW_EST[e](v) if len(v) != 0 else 0
                        per_key.setdefault(f"{ch}_Sharpwave_{e.title()}_{f}_{fname}", {})[pname] = val
        out = {}
        if self.s.apply_estimator_between_peaks_and_troughs:
            for ch in self.ch_names:
                for fname in self.filter_names:
                    for f, e in self.combos:
                        if f == "num_peaks":
                            continue
                        k = f"{ch}_Sharpwave_{e.title()}_{f}_{fname}"
                        vals = list(per_key[k].values())
                        out[k] = _SW_EST[e]([vals[0], vals[1]])
            if "num_peaks" in self.used:
                for ch in self.ch_names:
                    for fname in self.filter_names:
                        k = f"{ch}_Sharpwave_num_peaks_{fname}"
                        out[k] = (per_key[k]["Peak"] + per_key[k]["Trough"]) / 2
        else:
            for k, sub in per_key.items():
                for pname, v in sub.items():
                    out[f"{k}_analyze_{pname}"] = v
        return out


# --------------------------------------------------------------------------------------
# preprocessing (filter/notch_filter.py, processing/rereference.py, processing/resample.py)
# --------------------------------------------------------------------------------------


def notch_design(sfreq: float, line_noise, notch_width=3.0, trans_bandwidth: float = 6.8, freqs=None):
    """filter/notch_filter.py:25-76: band-stop bank at `freqs` (default k * line_noise), L = int(sfreq - 1);
    notch_width None -> freqs / 200, one width, or one per notch (:44-56)."""
    if freqs is None:
        freqs = np.arange(line_noise, sfreq / 2, line_noise, dtype=int)
    if freqs.size > 0 and freqs[-1] >= sfreq / 2:
        freqs = freqs[:-1]
    if freqs.size == 0:
        return None
    if notch_width is None:
        widths = freqs / 200.0
    elif np.any(np.asarray(notch_width) < 0):
        raise ValueError("notch_widths must be >= 0")
    else:
        widths = np.atleast_1d(notch_width)
        if len(widths) == 1:
            widths = widths[0] * np.ones_like(freqs)
        elif len(widths) != len(freqs):
            raise ValueError("notch_widths must be None, scalar, or the same length as freqs")
    tb_half = trans_bandwidth / 2.0
    lows = [f - w / 2.0 - tb_half for f, w in zip(freqs, widths)]
    highs = [f + w / 2.0 + tb_half for f, w in zip(freqs, widths)]
    return mne_restated.create_filter(None, sfreq, l_freq=highs, h_freq=lows,
                                      filter_length=int(sfreq - 1),
                                      l_trans_bandwidth=tb_half, h_trans_bandwidth=tb_half)


class NotchFilter:
    """filter/notch_filter.py:9-93."""

    def __init__(self, sfreq, line_noise=None, taps=None, freqs=None, notch_widths=3, trans_bandwidth=6.8) -> None:
        if line_noise is None and taps is None and freqs is None:
            raise ValueError("Either line_noise or freqs must be defined")
        self.taps = notch_design(sfreq, line_noise, notch_widths, trans_bandwidth, freqs) if taps is None else taps

    def process(self, data):
        if self.taps is None:
            return data
        return mne_restated._overlap_add_filter(data, self.taps)


def reref_matrix(names, rereference, used, types, status) -> np.ndarray | None:
    """processing/rereference.py:33-86 from plain column lists of the channel table."""
    idx = [i for i, u in enumerate(used) if u == 1]
    names = [names[i] for i in idx]
    refs = [rereference[i] for i in idx]
    types = [types[i] for i in idx]
    status = [status[i] for i in idx]
    n = len(names)
    if n in (0, 1):
        return None
    R = np.zeros((n, n))
    for i in range(n):
        R[i, i] = 1
        ref = refs[i]
        if ref is None or (isinstance(ref, float) and np.isnan(ref)) or \
                str(ref).lower() == "none" or status[i] != "good":
            continue
        if ref.lower() == "average":
            ridx = [j for j in range(n) if types[j] == types[i] and status[j] == "good" and j != i]
        else:
            ridx = []
            for rc in ref.split("&"):
                if rc not in names:
                    raise ValueError(f"One or more of the reference channels are not part of "
                                     f"the recording channels. First missing channel: {rc}.")
                if rc == names[i]:
                    raise ValueError(f"You cannot rereference to the same channel. Channel: {rc}.")
                ridx.append(names.index(rc))
        R[i, ridx] = -1 / len(ridx)
    good = [i for i in range(n) if status[i] == "good"]
    return R[np.ix_(good, good)]


class PreprocessingFilter:
    """processing/filter_preprocessing.py:44-94: single-range MNEFilter objects applied one after
    the other (each: zero-padded 'same' convolution, filter/mne_filter.py:82-128)."""

    def __init__(self, settings, sfreq, taps=None) -> None:
        pf = settings.preprocessing_filter
        if taps is None:
            enabled = pf.get_enabled()
            ranges = [tuple(getattr(pf, f"{n}_settings")) for n in enabled
                      if n not in ("lowpass_filter", "highpass_filter")]
            if "lowpass_filter" in enabled:
                ranges.append((None, pf.lowpass_filter_cutoff_hz))
            if "highpass_filter" in enabled:
                ranges.append((pf.highpass_filter_cutoff_hz, None))
            taps = []
            for lo, hi in ranges:
                try:
                    h = mne_restated.create_filter(None, sfreq, lo, hi, filter_length=int(sfreq - 1),
                                                   l_trans_bandwidth=4, h_trans_bandwidth=4)
                except ValueError:
                    h = mne_restated.create_filter(None, sfreq, lo, hi)
                taps.append(h)
        self.taps = list(taps)

    def process(self, data):
        for h in self.taps:
            data = fir_bank_apply(data, np.atleast_2d(h))[:, 0, :]
        return data


class Resampler:
    """processing/resample.py:19-60 (ratio 1 is a no-op; other ratios are parity-unpinned)."""

    def __init__(self, sfreq, resample_freq_hz) -> None:
        ratio = float(resample_freq_hz / sfreq)
        self.up = 0.0 if ratio == 1.0 else ratio

    def process(self, data):
        if not self.up:
            return data
        return mne_restated.resample(np.asarray(data, np.float64), up=self.up, down=1.0)


# --------------------------------------------------------------------------------------
# post-processing (processing/normalization.py) -- SURVEY 8(f) "next" #1
# --------------------------------------------------------------------------------------


# scikit-learn based methods (processing/normalization.py:57-70,166-186): every hop the reference FITS the
# scaler on nan_to_num(history) and transforms the current rows.  scikit-learn is a third-party dependency of the
# reference (pyproject: scikit-learn >= 1.x; 1.7.2 in this image); its published algorithms are restated here with
# NumPy and pinned by tests/golden/norm_methods.npz, which tests/golden/make_golden.py generates by running the
# reference's own Normalizer (with scikit-learn) in the build container:
#   RobustScaler        center = nanmedian, scale = nanpercentile 75 - 25 (linear), scale < 10 eps -> 1; (x - c) / s
#   MinMaxScaler        scale = 1 / (max - min) (range < 10 eps -> 1), min_ = -min * scale; x * scale + min_
#   QuantileTransformer(n_quantiles = 300, uniform): n_q = min(300, n_samples); quantiles = running maximum of
#       nanpercentile(history, linspace(0, 1, n_q) * 100); y = (interp(x, q, r) - interp(-x, -q[::-1], -r[::-1])) / 2,
#       x == q[0] -> 0, x == q[-1] -> 1.  Histories of more than 10 000 rows are randomly subsampled by
#       scikit-learn (subsample = 10 000, random_state = None): the reference itself is not reproducible there.
#   PowerTransformer    Yeo-Johnson, standardize=True: per column lambda = scipy.stats.yeojohnson_normmax (bounded Brent
#       search, scipy.optimize.fminbound, xtol 1.48e-8, on scipy's expm1 / log1p form of the log-likelihood, bounds
#       from the largest |x|), lambda = 1 for a constant column; the column is transformed with scikit-learn's
#       np.power form and standardised with the mean / variance of the transformed history (_yj_* below;
#       scikit-learn 1.7 with scipy >= 1.9, the versions of this image -- tests/test_oracle_golden.py checks the
#       restatement against PowerTransformer itself).
_SK_EPS10 = 10 * np.finfo(np.float64).eps
_F64 = np.finfo(np.float64)


def _yj_llf_neg(lmb, x, sl_sum):
    """-yeojohnson_llf(lmb, x) (scipy/stats/_morestats.py), +inf where the transformed variance underflows."""
    pos = x >= 0
    out = np.zeros_like(x)
    if abs(lmb) < np.spacing(1.0):
        out[pos] = np.log1p(x[pos])
    else:
        out[pos] = np.expm1(lmb * np.log1p(x[pos])) / lmb
    if abs(lmb - 2) > np.spacing(1.0):
        out[~pos] = -np.expm1((2 - lmb) * np.log1p(-x[~pos])) / (2 - lmb)
    else:
        out[~pos] = -np.log1p(-x[~pos])
    var = out.var()
    if var < _F64.tiny:
        return np.inf
    llf = -x.shape[0] / 2 * np.log(var) + (lmb - 1) * sl_sum
    return np.inf if np.isinf(llf) else -llf


def _fminbound(func, x1, x2, xatol=1.48e-8, maxfun=500):
    """scipy.optimize._optimize._minimize_scalar_bounded, statement for statement."""
    sqrt_eps = np.sqrt(2.2e-16)
    golden_mean = 0.5 * (3.0 - np.sqrt(5.0))
    a, b = x1, x2
    fulc = a + golden_mean * (b - a)
    nfc, xf = fulc, fulc
    rat = e = 0.0
    x = xf
    fx = func(x)
    num = 1
    ffulc = fnfc = fx
    xm = 0.5 * (a + b)
    tol1 = sqrt_eps * abs(xf) + xatol / 3.0
    tol2 = 2.0 * tol1
    while abs(xf - xm) > (tol2 - 0.5 * (b - a)):
        golden = 1
        if abs(e) > tol1:
            golden = 0
            r = (xf - nfc) * (fx - ffulc)
            q = (xf - fulc) * (fx - fnfc)
            p = (xf - fulc) * q - (xf - nfc) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = abs(q)
            r = e
            e = rat
            if (abs(p) < abs(0.5 * q * r)) and (p > q * (a - xf)) and (p < q * (b - xf)):
                rat = (p + 0.0) / q
                x = xf + rat
                if ((x - a) < tol2) or ((b - x) < tol2):
                    si = np.sign(xm - xf) + ((xm - xf) == 0)
                    rat = tol1 * si
            else:
                golden = 1
        if golden:
            e = a - xf if xf >= xm else b - xf
            rat = golden_mean * e
        si = np.sign(rat) + (rat == 0)
        x = xf + si * max(abs(rat), tol1)
        fu = func(x)
        num += 1
        if fu <= fx:
            if x >= xf:
                a = xf
            else:
                b = xf
            fulc, ffulc = nfc, fnfc
            nfc, fnfc = xf, fx
            xf, fx = x, fu
        else:
            if x < xf:
                a = x
            else:
                b = x
            if (fu <= fnfc) or (nfc == xf):
                fulc, ffulc = nfc, fnfc
                nfc, fnfc = x, fu
            elif (fu <= ffulc) or (fulc == xf) or (fulc == nfc):
                fulc, ffulc = x, fu
        xm = 0.5 * (a + b)
        tol1 = sqrt_eps * abs(xf) + xatol / 3.0
        tol2 = 2.0 * tol1
        if num >= maxfun:
            break
    return xf


def _yj_lambda(x):
    """scipy.stats.yeojohnson_normmax(x) for float64 data (brack=None): the bounded search."""
    if np.all(x == 0):
        return 1.0
    log1p_max_x = np.log1p(20 * np.max(np.abs(x)))
    log_eps = np.log(_F64.eps)
    lb = (np.log(_F64.tiny) - log_eps) / 2 / log1p_max_x
    ub = (np.log(_F64.max) + log_eps) / 2 / log1p_max_x
    if np.all(x < 0):
        lb, ub = 2 - ub, 2 - lb
    elif np.any(x < 0):
        lb, ub = max(2 - ub, lb), min(2 - lb, ub)
    sl_sum = (np.sign(x) * np.log1p(np.abs(x))).sum()
    return _fminbound(lambda l: _yj_llf_neg(l, x, sl_sum), lb, ub)


def yeo_johnson_conditioning(col, x_now, rel_noise=1e-13):
    """How far the standardised Yeo-Johnson output of `x_now` can move for likelihood-equivalent lambdas.  The
    log-likelihood is evaluated with a relative noise of ~rel_noise (float64 sums of n transcendental values; libm
    and summation order differ between NumPy and any other implementation); every lambda whose likelihood lies within
    that noise of the maximum is as good an answer as the one the bounded search happens to return.  Returns
    max |out(lambda) - out(lambda*)| over that interval (0 for a constant column)."""
    col = np.asarray(col, np.float64)
    n = col.shape[0]
    if _sk_constant(np.var(col), np.mean(col), n) or np.all(col == 0):
        return 0.0
    lmb = _yj_lambda(col)
    sl_sum = (np.sign(col) * np.log1p(np.abs(col))).sum()
    f0 = _yj_llf_neg(lmb, col, sl_sum)
    if not np.isfinite(f0):
        return np.inf
    # the noise of -llf = n/2 log(var) - ...: var itself carries rel_noise * (mean^2 / var + 1) from its cancellation
    with np.errstate(all="ignore"):
        t = _yj_transform(col, lmb)
    cond = 1.0 + np.mean(t) ** 2 / max(np.var(t), _F64.tiny)
    tol_f = rel_noise * cond * n / 2 + rel_noise * abs(f0)

    def out(l):
        with np.errstate(all="ignore"):
            tt = _yj_transform(col, l)
            m, v = np.mean(tt), np.var(tt)
            sc = 1.0 if _sk_constant(v, m, n) else np.sqrt(v)
            return float((_yj_transform(np.array([x_now], np.float64), l)[0] - m) / sc)

    o0, worst = out(lmb), 0.0
    for sign in (-1.0, 1.0):
        step = 1e-6
        while step < 64.0:
            l = lmb + sign * step
            f = _yj_llf_neg(l, col, sl_sum)
            if not np.isfinite(f) or f - f0 > tol_f:
                break
            o1 = out(l)
            if np.isfinite(o1):
                worst = max(worst, abs(o1 - o0))
            step *= 2.0
    return worst


def _yj_transform(x, lmb):
    """sklearn.preprocessing.PowerTransformer._yeo_johnson_transform (the np.power form)."""
    out = np.zeros_like(x)
    pos = x >= 0
    if abs(lmb) < np.spacing(1.0):
        out[pos] = np.log1p(x[pos])
    else:
        out[pos] = (np.power(x[pos] + 1, lmb) - 1) / lmb
    if abs(lmb - 2) > np.spacing(1.0):
        out[~pos] = -(np.power(-x[~pos] + 1, 2 - lmb) - 1) / (2 - lmb)
    else:
        out[~pos] = -np.log1p(-x[~pos])
    return out


def _sk_constant(var, mean, n):
    """sklearn.preprocessing._data._is_constant_feature."""
    eps = _F64.eps
    return var <= n * eps * var + (n * mean * eps) ** 2


def yeo_johnson_fit(col):
    """(lambda, mean, scale) of one history column the way PowerTransformer(standardize=True).fit does."""
    n = col.shape[0]
    lmb = 1.0 if _sk_constant(np.var(col), np.mean(col), n) else _yj_lambda(col)
    with np.errstate(all="ignore"):
        t = _yj_transform(col, lmb)
    mean, var = np.mean(t), np.var(t)
    scale = 1.0 if _sk_constant(var, mean, n) else np.sqrt(var)   # (_handle_zeros_in_scale with the constant mask)
    return lmb, mean, scale


def _sk_fit_transform(method, prev, cur, rng=None):
    """`rng`: numpy Generator for QuantileTransformer's random subsample of histories > 10 000 rows (the reference
    uses random_state=None: one realisation of a random variable; without `rng` such histories raise)."""
    X = np.nan_to_num(np.asarray(prev, np.float64))
    cur = np.array(cur, dtype=np.float64)
    one_d = cur.ndim == 1
    Y = cur[None] if one_d else cur
    with np.errstate(invalid="ignore", divide="ignore"):
        if method == "robust":
            center = np.nanmedian(X, axis=0)
            q = np.nanpercentile(X, (25.0, 75.0), axis=0)
            scale = q[1] - q[0]
            scale[scale < _SK_EPS10] = 1.0
            out = (Y - center) / scale
        elif method == "minmax":
            lo, hi = np.nanmin(X, axis=0), np.nanmax(X, axis=0)
            rng = hi - lo
            rng[rng < _SK_EPS10] = 1.0
            scale = 1.0 / rng
            out = Y * scale + (0.0 - lo * scale)
        elif method == "quantile":
            n = X.shape[0]
            if n > 10000:
                if rng is None:
                    raise NotImplementedError("QuantileTransformer subsamples histories of more than 10 000 rows at random")
                X = X[rng.choice(n, 10000, replace=False)]   # sklearn.utils.resample(replace=False): rows, jointly
                n = 10000
            nq = min(300, n)
            refs = np.linspace(0, 1, nq, endpoint=True)
            quant = np.maximum.accumulate(np.nanpercentile(X, refs * 100, axis=0))
            out = np.empty_like(Y)
            for j in range(Y.shape[1]):
                col, qj = Y[:, j].copy(), quant[:, j]
                lower, upper = col == qj[0], col == qj[-1]
                fin = ~np.isnan(col)
                cf = col[fin]
                col[fin] = 0.5 * (np.interp(cf, qj, refs) - np.interp(-cf, -qj[::-1], -refs[::-1]))
                col[upper] = 1.0
                col[lower] = 0.0
                out[:, j] = col
        elif method == "power":
            out = np.empty_like(Y)
            for j in range(Y.shape[1]):
                lmb, mean, scale = yeo_johnson_fit(X[:, j].copy())
                out[:, j] = (_yj_transform(Y[:, j].copy(), lmb) - mean) / scale
        else:
            raise NotImplementedError(method)
    return out[0] if one_d else out


_SK_METHODS = ("robust", "minmax", "quantile", "power")


class RawNormalizer:
    """processing/normalization.py:31-116 with type "raw": history = the first window plus the last
    int(sfreq / feat_hz) samples of every later window, statistics over it incl. the current tail,
    trimmed to N - 1 = int(normalization_time_s * sfreq) - 1 samples afterwards."""

    def __init__(self, sfreq, settings, rng=None) -> None:
        rs = settings.raw_normalization_settings
        self.rng = rng
        self.method = rs.normalization_method
        self.clip = rs.clip
        self.add = int(sfreq / settings.sampling_rate_features_hz)
        self.n = int(rs.normalization_time_s * sfreq)
        self.prev = np.empty((0, 0))

    def process(self, data):
        if self.prev.size == 0:
            self.prev = data.T
            return data
        cur = data.T
        self.prev = np.vstack((self.prev, cur[-self.add:]))
        if self.method in _SK_METHODS:
            out = _sk_fit_transform(self.method, self.prev, cur, self.rng)
        else:
            has_nan = np.any(np.isnan(sum(self.prev)))
            mean = (np.nanmean if has_nan else np.mean)(self.prev, axis=0)
            std = (np.nanstd if has_nan else np.std)(self.prev, axis=0)
            std[std == 0] = 1
            with np.errstate(divide="ignore", invalid="ignore"):
                if self.method == "mean":
                    out = (cur - mean) / mean
                elif self.method == "zscore":
                    out = (cur - mean) / std
                elif self.method in ("median", "zscore-median"):
                    med = (np.nanmedian if has_nan else np.median)(self.prev, axis=0)
                    out = (cur - med) / (med if self.method == "median" else std)
                else:
                    raise NotImplementedError(self.method)
        if self.clip:
            out = out.clip(min=-self.clip, max=self.clip)
        self.prev = self.prev[-self.n + 1:]
        return np.nan_to_num(out).T


class FeatureNormalizer:
    """processing/normalization.py:31-111 for mean / median / zscore / zscore-median / robust / minmax / quantile."""

    def __init__(self, settings) -> None:
        s = settings.feature_normalization_settings
        self.method = s.normalization_method
        self.clip = s.clip
        self.n = int(s.normalization_time_s * settings.sampling_rate_features_hz)
        self.prev = np.empty((0, 0))

    def process(self, cur: np.ndarray) -> np.ndarray:
        if self.prev.size == 0:
            self.prev = cur
            return cur
        self.prev = np.vstack((self.prev, cur))
        if self.method in _SK_METHODS:
            out = _sk_fit_transform(self.method, self.prev, cur)
            if self.clip:
                out = out.clip(min=-self.clip, max=self.clip)
            self.prev = self.prev[-self.n + 1:]
            return np.nan_to_num(out)
        has_nan = np.any(np.isnan(sum(self.prev)))
        mean = (np.nanmean if has_nan else np.mean)(self.prev, axis=0)
        med = (np.nanmedian if has_nan else np.median)(self.prev, axis=0)
        std = (np.nanstd if has_nan else np.std)(self.prev, axis=0)
        std[std == 0] = 1
        with np.errstate(divide="ignore", invalid="ignore"):
            if self.method == "mean":
                out = (cur - mean) / mean
            elif self.method == "median":
                out = (cur - med) / med
            elif self.method == "zscore":
                out = (cur - mean) / std
            elif self.method == "zscore-median":
                out = (cur - med) / std
            else:
                raise NotImplementedError(self.method)
        if self.clip:
            out = out.clip(min=-self.clip, max=self.clip)
        self.prev = self.prev[-self.n + 1:]
        return np.nan_to_num(out)


# --------------------------------------------------------------------------------------
# orchestrator (stream/data_processor.py:238-311, features/feature_processor.py:45-84)
# --------------------------------------------------------------------------------------

FEATURE_ORDER = ["raw_hjorth", "return_raw", "bandpass_filter", "stft", "fft", "welch",
                 "sharpwave_analysis", "fooof", "nolds", "coherence", "bursts", "linelength",
                 "mne_connectivity", "bispectrum"]
_FEATURE_CLS = {"raw_hjorth": Hjorth, "return_raw": Raw, "bandpass_filter": BandPower,
                "stft": STFT, "fft": FFT, "welch": Welch, "sharpwave_analysis": SharpwaveAnalyzer,
                "bursts": Bursts, "linelength": LineLength}
_PREPROC_ORDER = ["preprocessing_filter", "notch_filter", "raw_resampling", "re_referencing",
                  "raw_normalization"]


class DataProcessor:
    """stream/data_processor.py:19-311 restricted to the SURVEY section-8 scope.

    ``channels`` is a mapping of column name -> list (name, rereference, used, target,
    type, status, new_name), e.g. ``df.to_dict("list")``.
    """

    def __init__(self, sfreq, settings, channels: dict, line_noise=None) -> None:
        self.settings = settings
        self.sfreq = sfreq // 1
        ch = channels
        n = len(ch["name"])
        self.ch_names_used = [ch["new_name"][i] for i in range(n)
                              if ch["used"][i] == 1 and ch["status"][i] == "good"]
        self.feature_idx = [i for i in range(n) if ch["used"][i] and not ch["target"][i]
                            and ch["status"][i] == "good"]
        self.pre = []
        for name in _PREPROC_ORDER:
            if name not in settings.preprocessing:
                continue
            if name == "notch_filter":
                self.pre.append(NotchFilter(self.sfreq, line_noise))
            elif name == "raw_resampling":
                self.pre.append(Resampler(self.sfreq, settings.raw_resampling_settings.resample_freq_hz))
            elif name == "re_referencing":
                R = reref_matrix(ch["name"], ch["rereference"], ch["used"], ch["type"], ch["status"])
                self.pre.append(_Reref(R))
            elif name == "preprocessing_filter":
                self.pre.append(PreprocessingFilter(settings, self.sfreq))
            elif name == "raw_normalization":
                self.pre.append(RawNormalizer(self.sfreq, settings))
            else:
                raise NotImplementedError(f"{name} is out of scope (SURVEY.md section 2)")
        self.features = []
        for f in _enabled(settings.features):
            if f not in _FEATURE_CLS:
                raise NotImplementedError(f"feature {f} is out of scope (SURVEY.md section 2)")
            self.features.append(_FEATURE_CLS[f](settings, self.ch_names_used, self.sfreq))
        self.normalizer = (FeatureNormalizer(settings)
                           if settings.postprocessing.feature_normalization else None)
        self.non_psd = None

    def preprocess(self, data):
        for p in self.pre:
            data = p.process(data)
        return data

    def process(self, data: np.ndarray) -> dict:
        nan_channels = np.isnan(data).any(axis=1)
        data = np.nan_to_num(data)[self.feature_idx, :]
        data = self.preprocess(data)
        feats: dict = {}
        for f in self.features:
            feats.update(f.calc_feature(data))
        if self.normalizer is not None:
            keys = list(feats.keys())
            vals = np.fromiter(feats.values(), dtype=np.float64)
            if not self.settings.feature_normalization_settings.normalize_psd:
                if self.non_psd is None:
                    self.non_psd = [i for i, k in enumerate(keys) if "psd" not in k]
                normed = vals.copy()
                normed[self.non_psd] = self.normalizer.process(vals[self.non_psd])
            else:
                normed = self.normalizer.process(vals)
            feats = dict(zip(keys, normed))
        if nan_channels.sum() > 0:
            for ch in list(np.array(self.ch_names_used)[nan_channels]):
                for k in feats:
                    if ch in k:
                        feats[k] = np.nan
        return feats


class _Reref:
    def __init__(self, R):
        self.R = R

    def process(self, data):
        return data if self.R is None else self.R @ data


def run_stream(data: np.ndarray, sfreq: float, settings, channels: dict, line_noise=50):
    """stream/stream.py:198-345 restricted to feature computation: list of per-window dicts
    (with ``time`` and target columns), in reference column order."""
    dp = DataProcessor(sfreq, settings, channels, line_noise)
    starts, ends, times = window_schedule(data.shape[1], sfreq, settings.sampling_rate_features_hz,
                                          settings.segment_length_features_ms)
    tidx = [i for i, t in enumerate(channels["target"]) if t == 1]
    rows = []
    for s, e, t in zip(starts, ends, times):
        w = data[:, s:e]
        d = dp.process(w)
        d["time"] = t
        for i in tidx:
            d[channels["name"][i]] = w[i, -1]
        rows.append({k: float(v) for k, v in d.items()})
    return rows


# --------------------------------------------------------------------------------------
# Conditioning reports (test infrastructure for tests/parity.py): the engine computes in fp32,
# the reference in float64.  Three families of outputs are NOT Lipschitz in the data -- log10 of a
# spectral magnitude near a null, find_peaks' discrete selections, the `env >= thr` comparison of
# the burst detector -- so a tolerance miss there is only accepted when the float64 computation
# itself shows the ill-conditioning.  These functions report it per output; nothing here is part
# of the restated arithmetic.
# --------------------------------------------------------------------------------------


def spectral_magnitudes(family: str, settings, sfreq: float, x_row: np.ndarray):
    """Linear magnitudes behind one channel's FFT / Welch / STFT features:
    (mag[n_freq] or mag[n_freq, n_seg], [(band, bin_indices)], freqs, white_gain).
    Welch returns sqrt(PSD) so that all three are amplitudes.  ``white_gain`` = the rms magnitude a
    white input of unit rms produces in this family (FFT: sqrt(N); Welch: sqrt(2 / fs) with density
    scaling and one-sided doubling; STFT: sqrt(sum w^2) / sum w with spectrum scaling)."""
    x = np.asarray(x_row, np.float64)[None]
    if family == "fft":
        o = FFT(settings, ["c"], sfreq)
        mag = np.abs(sp_fft.rfft(x[:, -o.N:], axis=-1))[0]
        gain = math.sqrt(o.N)
    elif family == "welch":
        o = Welch(settings, ["c"], sfreq)
        mag = np.sqrt(welch_psd(x, o.sfreq, o.sfreq))[0]
        gain = math.sqrt(2.0 / o.sfreq)
    elif family == "stft":
        o = STFT(settings, ["c"], sfreq)
        n = min(o.nperseg, x.shape[-1])   # scipy's nperseg clamp (STFT.spectrum)
        mag = stft_mag(x, n)[0]
        w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n) / n)
        gain = math.sqrt(np.sum(w * w)) / np.sum(w)
    else:
        raise ValueError(family)
    return mag, o.idx_range, o.freqs, gain


def spectral_null_ratio(mag: np.ndarray, idx: np.ndarray, floor_rms: float) -> float:
    """min |X_k| over the contributing bins / ``floor_rms``, the magnitude white noise with the rms of
    the WINDOW (DC offset included) would have in this family (white_gain * rms(x)).
    Every fp32 sample of the window -- after the fp32 pre-processing -- carries a rounding error
    relative to its own size, DC included (a 500 uV offset on a 50 uV signal costs a decade); as white
    noise it adds ~6e-8 * floor_rms to every bin of every family, also those (Welch: detrended, STFT:
    windowed) whose own spectrum no longer shows the DC.  log10 turns that absolute error into a relative
    one: a contributing bin at ratio r carries ~1e-7 / r in log10 units."""
    m = np.asarray(mag, np.float64)
    sel = m[np.asarray(idx, dtype=int)]
    if sel.size == 0 or floor_rms == 0.0:
        return float("inf")
    return float(sel.min() / floor_rms)


def spectral_log_error_bound(mag: np.ndarray, idx: np.ndarray, floor_rms: float, eps: float,
                             power: bool, estimator: str) -> float:
    """Largest change of the log10-valued entry that an absolute error of eps * floor_rms on every
    contributing bin explains: per bin log10(1 + eps * floor_rms / |X_k|) (twice that for a power
    spectrum); their MEAN for the "mean" estimator, their MAX for median / std / max and single-bin
    ("psd") entries (one bin can move those by at most its own error)."""
    m = np.asarray(mag, np.float64)
    sel = np.abs(m[np.asarray(idx, dtype=int)])
    if sel.size == 0:
        return 0.0
    e = np.asarray(eps, np.float64)          # scalar, or one value per contributing bin
    if e.ndim == 1 and sel.ndim == 2:
        e = e[:, None]
    with np.errstate(divide="ignore"):
        per = np.log10(1.0 + e * floor_rms / sel) * (2.0 if power else 1.0)
    return float(per.mean() if estimator == "mean" else per.max())


def _strict_extrema(z: np.ndarray) -> np.ndarray:
    return find_peaks_distance(z, None)


def sharpwave_decision_margin(y: np.ndarray, dist_peaks: float, dist_troughs: float) -> float:
    """Smallest absolute perturbation of the filtered series ``y`` that can change a discrete
    decision of features/sharpwaves.py:339-374 (both polarities):
      * existence / position of a local extremum (find_peaks compares neighbouring samples):
        min_i |y[i+1] - y[i]|;
      * find_peaks' distance suppression keeps the HIGHER of two extrema closer than `distance`:
        min |y[p] - y[q]| over same-kind extrema p, q with |p - q| < ceil(distance).
    The trough/peak pairing that follows is index arithmetic on these sets (no further comparisons
    of values)."""
    y = np.asarray(y, np.float64)
    m = float(np.min(np.abs(np.diff(y)))) if y.size > 1 else float("inf")
    d = int(math.ceil(max(dist_peaks, dist_troughs)))
    for z in (y, -y):
        p = _strict_extrema(z)
        for lag in range(1, min(d, p.size)):
            a, b = p[:-lag], p[lag:]
            close = (b - a) < d
            if not close.any():
                break   # (sorted positions: larger lags are farther apart)
            m = min(m, float(np.min(np.abs(z[a[close]] - z[b[close]]))))
    return m


def burst_decision_margin(env: np.ndarray, thr: float) -> float:
    """min_n |env[n] - thr|: the perturbation that flips one `env >= thr` sample
    (features/bursts.py:175); a flipped sample moves run lengths / counts by whole samples."""
    return float(np.min(np.abs(np.asarray(env, np.float64) - float(thr))))


def hjorth_noise_bound(y: np.ndarray, sigma: float):
    """Relative change of Hjorth mobility / complexity (features/hjorth_raw.py:24-34, bandpower.py:185-207) that white
    noise of standard deviation `sigma` on the samples of `y` explains.  The three variances are v_k = var(diff^k y);
    noise adds c_k sigma^2 (c = 1, 2, 6: the squared binomial weights) to v_k plus a cross term whose 3-sigma size is
    6 sqrt(c_k sigma^2 v_k / N).  A series sampled far above its band (theta at 2 kHz) has v_2 << v_0, so its
    complexity = sqrt(v_2 v_0) / v_1 amplifies sample noise by (fs / f)^2; a band in the stop band of a pre-processing
    filter has v_0 itself far below the input power.  Returns (activity, mobility, complexity) bounds."""
    y = np.asarray(y, np.float64)
    n = max(y.shape[-1] - 2, 1)
    d1 = np.diff(y)
    v = [float(np.var(y)), float(np.var(d1)), float(np.var(np.diff(d1)))]
    rel = []
    for c, vk in zip((1.0, 2.0, 6.0), v):
        rel.append((c * sigma * sigma + 6.0 * math.sqrt(c * sigma * sigma * vk / n)) / vk if vk > 0 else np.inf)
    return rel[0], 0.5 * (rel[0] + rel[1]), 0.5 * (rel[0] + rel[2]) + rel[1]

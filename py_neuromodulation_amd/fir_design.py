"""Host-side FIR tap design (run once per stream; the taps are an input of the HIP kernels).

The reference delegates design to MNE-Python, which is neither vendored nor installable
here ("parity unpinned" for the design step).  This module implements the
published MNE algorithm with NumPy only -- windowed-sinc low-pass sections (hamming,
length 3.3 / transition), added/subtracted per gain change -- for the three call shapes on
the hot path:

  band_pass_bank   filter/mne_filter.py:35-80    (filter_length = int(sfreq - 1), 4 Hz
                                                   transitions, auto-length fallback)
  auto_band_pass   features/sharpwaves.py:127-143 (auto transitions and length)
  notch_bank       filter/notch_filter.py:25-76   (band-stops at k * line_noise)
  fir_filter / preprocessing_filter_bank   processing/filter_preprocessing.py:44-79 (single low-,
                                                   high-, band-pass or band-stop filters)
"""

from __future__ import annotations

import functools
import math

import numpy as np

_HAMMING_LENGTH_FACTOR = 3.3


@functools.lru_cache(maxsize=512)
def _hamming_lowpass(numtaps: int, cutoff: float) -> np.ndarray:
    """Windowed-sinc low-pass, cutoff as a fraction of Nyquist, unit DC gain
    (== scipy.signal.firwin(numtaps, cutoff, window="hamming", fs=2)).  Read-only and cached: every plan of a process with
    the same bands designs the same ~26 sections (1 ms of a fresh Stream's construction)."""
    m = np.arange(numtaps) - 0.5 * (numtaps - 1)
    h = cutoff * np.sinc(cutoff * m)
    if numtaps > 1:
        h = h * (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(numtaps) / (numtaps - 1)))
    h = h / h.sum()
    h.flags.writeable = False
    return h


def _sections(n_taps: int, edges: np.ndarray, gains: np.ndarray) -> np.ndarray:
    """Sum of low-pass sections for a piecewise 0/1 gain profile (edges normalised to Nyquist)."""
    if n_taps % 2 == 0:
        raise RuntimeError("zero-phase FIR needs an odd number of taps")
    h = np.zeros(n_taps)
    if gains[-1] == 1:
        h[n_taps // 2] = 1.0
    hi_f, hi_g = edges[-1], gains[-1]
    for lo_f, lo_g in zip(edges[-2::-1], gains[-2::-1]):
        if lo_g != hi_g:
            width = (hi_f - lo_f) / 2.0
            n = int(round(_HAMMING_LENGTH_FACTOR / width))
            n += 1 - n % 2
            if n > n_taps:
                raise ValueError(f"filter length {n_taps} too short for the transition band "
                                 f"(needs {n} taps)")
            sec = _hamming_lowpass(n, (hi_f + lo_f) / 2.0)
            pad = (n_taps - n) // 2
            if lo_g == 0:
                h[pad:n_taps - pad] -= sec
            else:
                h[pad:n_taps - pad] += sec
        hi_f, hi_g = lo_f, lo_g
    return h


def _odd(n: int) -> int:
    return n + (n - 1) % 2


def _auto_length(sfreq: float, *transitions: float) -> int:
    seconds = _HAMMING_LENGTH_FACTOR / min(transitions)
    # MNE formats the duration as "%ss" and parses it back before ceil()
    return _odd(max(int(math.ceil(float("%s" % (seconds,)) * sfreq)), 1))


def band_pass(sfreq: float, l_freq: float, h_freq: float, filter_length: int | None = None,
              l_trans: float | None = None, h_trans: float | None = None) -> np.ndarray:
    """Band-pass taps; ``None`` transitions / length mean MNE's "auto"."""
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    if h_freq > nyq:
        raise ValueError(f"h_freq ({h_freq}) must be below the Nyquist frequency {nyq}")
    if not l_freq < h_freq:
        raise ValueError("band-pass needs l_freq < h_freq")
    lt = min(max(0.25 * l_freq, 2.0), l_freq) if l_trans is None else float(l_trans)
    ht = min(max(0.25 * h_freq, 2.0), nyq - h_freq) if h_trans is None else float(h_trans)
    if lt <= 0 or ht <= 0:
        raise ValueError("transition bandwidths must be positive")
    s1, s2 = l_freq - lt, h_freq + ht
    if s1 < 0:
        raise ValueError("Filter specification invalid: lower stop frequency negative")
    if s2 > nyq:
        raise ValueError("Effective band-stop frequency is too high")
    n = _auto_length(sfreq, lt, ht) if filter_length is None else _odd(int(filter_length))
    edges, gains = [s1, l_freq, h_freq, s2], [0, 1, 1, 0]
    if s2 != nyq:
        edges.append(nyq)
        gains.append(0)
    if s1 != 0:
        edges.insert(0, 0.0)
        gains.insert(0, 0)
    return _sections(n, np.asarray(edges) / nyq, np.asarray(gains))


def band_pass_bank(f_ranges, sfreq: float, filter_length: float | None = None,
                   l_trans: float = 4, h_trans: float = 4) -> np.ndarray:
    """MNEFilter.__init__: one row per band, auto-length fallback when the fixed length is too
    short for the transition band.  Rows must share a length (np.vstack in the reference)."""
    if filter_length is None:
        filter_length = sfreq - 1
    rows = []
    for lo, hi in f_ranges:
        try:
            rows.append(band_pass(sfreq, lo, hi, int(filter_length), l_trans, h_trans))
        except ValueError:
            rows.append(band_pass(sfreq, lo, hi))
    return np.vstack(rows)


def notch_bank(sfreq: float, line_noise: float | None, notch_width=3.0,
               trans_bandwidth: float = 6.8, freqs=None) -> np.ndarray | None:
    """NotchFilter.__init__ (filter/notch_filter.py:9-76): multi band-stop, length int(sfreq - 1), at the explicit
    ``freqs`` or at k * line_noise; ``notch_width``: None (= freqs / 200), one width, or one per notch."""
    if freqs is None:
        if line_noise is None:
            raise ValueError("Either line_noise or freqs must be defined if notch_filter is activated.")
        freqs = np.arange(line_noise, sfreq / 2, line_noise, dtype=int)
    else:
        freqs = np.atleast_1d(np.asarray(freqs))
    if freqs.size > 0 and freqs[-1] >= sfreq / 2:
        freqs = freqs[:-1]
    if freqs.size == 0:
        return None
    if notch_width is None:
        notch_width = freqs / 200.0
    elif np.any(np.asarray(notch_width) < 0):
        raise ValueError("notch_widths must be >= 0")
    else:
        notch_width = np.atleast_1d(notch_width)
        if len(notch_width) == 1:
            notch_width = notch_width[0] * np.ones_like(freqs)   # (integer freqs x integer width stay integers, :50)
        elif len(notch_width) != len(freqs):
            raise ValueError("notch_widths must be None, scalar, or the same length as freqs")
    freqs = np.asarray(freqs, dtype=np.float64)
    notch_width = np.asarray(notch_width, dtype=np.float64)
    nyq = float(sfreq) / 2.0
    tb = trans_bandwidth / 2.0
    lows = freqs - notch_width / 2.0 - tb     # pass-band edges below each notch
    highs = freqs + notch_width / 2.0 + tb    # pass-band edges above each notch
    if np.any(lows < 0):
        raise ValueError("Filter specification invalid: lower stop frequency negative")
    if np.any(highs > nyq):
        raise ValueError("Effective band-stop frequency is too high")
    edges = np.r_[lows, lows + tb, highs - tb, highs]
    gains = np.r_[np.ones_like(lows), np.zeros_like(lows), np.zeros_like(lows), np.ones_like(lows)]
    order = np.argsort(edges)
    edges, gains = edges[order], gains[order]
    if edges[0] != 0:
        edges, gains = np.r_[0.0, edges], np.r_[1.0, gains]
    if edges[-1] != nyq:
        edges, gains = np.r_[edges, nyq], np.r_[gains, 1.0]
    if np.any(np.abs(np.diff(gains, 2)) > 1):
        raise ValueError("Stop bands are not sufficiently separated.")
    return _sections(_odd(int(sfreq - 1)), edges / nyq, gains)


def fir_filter(sfreq: float, l_freq: float | None, h_freq: float | None, filter_length: int | None = None,
               l_trans: float | None = None, h_trans: float | None = None) -> np.ndarray:
    """One zero-phase hamming FIR the way ``mne.filter.create_filter`` picks the response:
    ``l_freq is None`` low-pass, ``h_freq is None`` high-pass, ``l_freq < h_freq`` band-pass,
    ``l_freq > h_freq`` band-stop.  ``None`` transitions / length mean MNE's "auto"."""
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    if l_freq is not None and float(l_freq) == 0.0:
        l_freq = None
    if l_freq is None and h_freq is None:
        raise ValueError("all-pass filter requested")
    if h_freq is not None and h_freq > nyq:
        raise ValueError(f"h_freq ({h_freq}) must be below the Nyquist frequency {nyq}")

    def auto_l(f):
        return min(max(0.25 * f, 2.0), f)

    def auto_h(f):
        return min(max(0.25 * f, 2.0), nyq - f)

    if l_freq is None:                       # low-pass
        ht = auto_h(h_freq) if h_trans is None else float(h_trans)
        if ht <= 0:
            raise ValueError("transition bandwidths must be positive")
        f_s = h_freq + ht
        if f_s > nyq:
            raise ValueError("Effective stop frequency too high")
        n = _auto_length(sfreq, ht) if filter_length is None else _odd(int(filter_length))
        edges, gains = [0.0, h_freq, f_s], [1, 1, 0]
        if f_s != nyq:
            edges.append(nyq)
            gains.append(0)
    elif h_freq is None:                     # high-pass
        lt = auto_l(l_freq) if l_trans is None else float(l_trans)
        if lt <= 0:
            raise ValueError("transition bandwidths must be positive")
        f_s = l_freq - lt
        if f_s < 0:
            raise ValueError("Filter specification invalid: lower stop frequency negative")
        n = _auto_length(sfreq, lt) if filter_length is None else _odd(int(filter_length))
        edges, gains = [f_s, l_freq, nyq], [0, 1, 1]
        if f_s != 0:
            edges.insert(0, 0.0)
            gains.insert(0, 0)
    elif l_freq < h_freq:                    # band-pass
        return band_pass(sfreq, l_freq, h_freq, filter_length, l_trans, h_trans)
    else:                                    # band-stop: pass below h_freq and above l_freq
        lo, hi = float(h_freq), float(l_freq)
        lt = auto_l(lo) if h_trans is None else float(h_trans)   # MNE swaps the roles (reverse=True)
        ht = auto_h(hi) if l_trans is None else float(l_trans)
        if lt <= 0 or ht <= 0:
            raise ValueError("transition bandwidths must be positive")
        if lo < 0:
            raise ValueError("Filter specification invalid: lower stop frequency negative")
        if hi > nyq:
            raise ValueError("Effective band-stop frequency is too high")
        n = _auto_length(sfreq, lt, ht) if filter_length is None else _odd(int(filter_length))
        edges, gains = [lo, lo + lt, hi - ht, hi], [1, 0, 0, 1]
        if edges[0] != 0:
            edges.insert(0, 0.0)
            gains.insert(0, 1)
        if edges[-1] != nyq:
            edges.append(nyq)
            gains.append(1)
    return _sections(n, np.asarray(edges, dtype=float) / nyq, np.asarray(gains))


def mne_filter_single(sfreq: float, l_freq, h_freq, filter_length: float | None = None) -> np.ndarray:
    """MNEFilter.__init__ for ONE range (filter/mne_filter.py:51-76): 4 Hz transitions and
    int(filter_length) taps; on ValueError the same range with MNE's automatic transitions/length."""
    if filter_length is None:
        filter_length = sfreq - 1
    try:
        return fir_filter(sfreq, l_freq, h_freq, int(filter_length), 4.0, 4.0)
    except ValueError:
        return fir_filter(sfreq, l_freq, h_freq)


def preprocessing_filter_bank(pf_settings, sfreq: float) -> list[np.ndarray]:
    """Taps of PreprocessingFilter in application order (processing/filter_preprocessing.py:50-79):
    the enabled ones of (bandstop_filter, bandpass_filter) in selector order -- their [lo, hi] range is
    handed to create_filter as (l_freq, h_freq), so the default "bandstop" [100, 160] is in fact a
    band-PASS, as in the reference -- then the low-pass, then the high-pass."""
    enabled = pf_settings.get_enabled()
    taps = []
    for name in enabled:
        if name in ("lowpass_filter", "highpass_filter"):
            continue
        rng = getattr(pf_settings, f"{name}_settings")
        taps.append(mne_filter_single(sfreq, float(rng[0]), float(rng[1])))
    if "lowpass_filter" in enabled:
        taps.append(mne_filter_single(sfreq, None, float(pf_settings.lowpass_filter_cutoff_hz)))
    if "highpass_filter" in enabled:
        taps.append(mne_filter_single(sfreq, float(pf_settings.highpass_filter_cutoff_hz), None))
    return taps

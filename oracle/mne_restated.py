"""Restatement of the three MNE-Python functions the reference's hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: ``mne`` is an
unpinned dependency of the reference (pyproject.toml:36,47), it is not vendored under
/root/reference and not installable in this image.  The algorithms below restate MNE's
published ``mne/filter.py`` (BSD-3) -- ``create_filter`` / ``_triage_filter_params`` /
``_firwin_design`` / ``_overlap_add_filter`` / ``_smart_pad`` / ``resample`` -- and are
anchored on the reference's call sites:

  * filter/mne_filter.py:53-76      create_filter(None, sfreq, l, h, fir_design="firwin",
                                     l/h_trans_bandwidth=4, filter_length=int(sfreq-1)),
                                     fallback to auto length on ValueError
  * filter/notch_filter.py:62-76    create_filter(l_freq=highs, h_freq=lows, ...) band-stop bank
  * filter/notch_filter.py:84-93    _overlap_add_filter(x, h, phase="zero", pad="reflect_limited")
  * features/sharpwaves.py:127-143  create_filter(None, sfreq, l, h, fir_design="firwin") (auto)
  * processing/resample.py:58-60    resample(x.astype(f64), up=ratio, down=1.0)

The FIR window is always hamming on those paths (MNE default ``fir_window``).
"""

from __future__ import annotations

import numpy as np

_LENGTH_FACTOR_HAMMING = 3.3  # MNE _length_factors["hamming"]


def _firwin_lowpass(numtaps: int, cutoff_norm: float) -> np.ndarray:
    """scipy.signal.firwin(numtaps, cutoff, window="hamming", pass_zero=True, fs=2).

    Kept as a direct SciPy call: the reference's own dependency does exactly this.
    """
    from scipy.signal import firwin

    return firwin(numtaps, cutoff_norm, window="hamming", pass_zero=True, fs=2.0)


def _firwin_design(N: int, freq: np.ndarray, gain: np.ndarray) -> np.ndarray:
    """MNE ``_firwin_design``: sum/difference of windowed-sinc low-passes.

    ``freq`` is normalised to Nyquist = 1 and starts at 0; gains are 0/1.
    Raises ValueError when a transition needs more taps than ``N`` (this is the
    exception the reference catches at filter/mne_filter.py:64).
    """
    assert freq[0] == 0 and len(freq) == len(gain) and N % 2 == 1
    h = np.zeros(N)
    prev_freq = freq[-1]
    prev_gain = gain[-1]
    if gain[-1] == 1:
        h[N // 2] = 1.0  # start with "all up"
    for this_freq, this_gain in zip(freq[::-1][1:], gain[::-1][1:]):
        if this_gain != prev_gain:
            transition = (prev_freq - this_freq) / 2.0
            this_N = int(round(_LENGTH_FACTOR_HAMMING / transition))
            this_N += 1 - this_N % 2  # make it odd
            if this_N > N:
                raise ValueError(
                    f"The requested filter length {N} is too short for the requested "
                    f"transition band, which requires {this_N} samples"
                )
            this_h = _firwin_lowpass(this_N, (prev_freq + this_freq) / 2.0)
            offset = (N - this_N) // 2
            if this_gain == 0:
                h[offset : N - offset] -= this_h
            else:
                h[offset : N - offset] += this_h
        prev_gain = this_gain
        prev_freq = this_freq
    return h


def _auto_trans(l_freq, h_freq, sfreq):
    lt = None if l_freq is None else np.minimum(np.maximum(0.25 * l_freq, 2.0), l_freq)
    ht = (
        None
        if h_freq is None
        else np.minimum(np.maximum(0.25 * h_freq, 2.0), sfreq / 2.0 - h_freq)
    )
    return lt, ht


def _to_samples(filter_length, sfreq: float) -> int:
    """MNE ``_to_samples`` + the odd-length rule for firwin / zero phase."""
    if isinstance(filter_length, str):
        low = filter_length.lower()
        if low.endswith("ms"):
            mult, num = 1e-3, low[:-2]
        elif low.endswith("s"):
            mult, num = 1.0, low[:-1]
        else:
            raise ValueError(f"filter_length string must end in 's' or 'ms': {filter_length}")
        n = max(int(np.ceil(float(num) * mult * sfreq)), 1)
    else:
        n = int(filter_length)
    n += (n - 1) % 2
    return n


def create_filter(
    data,
    sfreq,
    l_freq,
    h_freq,
    filter_length="auto",
    l_trans_bandwidth="auto",
    h_trans_bandwidth="auto",
    method="fir",
    iir_params=None,
    phase="zero",
    fir_window="hamming",
    fir_design="firwin",
    verbose=None,
):
    """Restated ``mne.filter.create_filter`` for method="fir", firwin, hamming, zero phase."""
    if method != "fir" or fir_design != "firwin" or fir_window != "hamming" or phase != "zero":
        raise NotImplementedError("only the FIR/firwin/hamming/zero-phase path is restated")
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    if h_freq is not None:
        h_freq = np.array(h_freq, float).ravel()
        if (h_freq > nyq).any():
            raise ValueError(f"h_freq ({h_freq}) must be less than the Nyquist frequency {nyq}")
    if l_freq is not None:
        l_freq = np.array(l_freq, float).ravel()
        if (l_freq == 0).all():
            l_freq = None
    if l_freq is None and h_freq is None:
        raise NotImplementedError("all-pass not used on the hot path")

    def _resolve_trans(freq_, trans_, which):
        if isinstance(trans_, str):
            if trans_ != "auto":
                raise ValueError("trans_bandwidth must be 'auto' if string")
            lt, ht = _auto_trans(freq_ if which == "l" else None,
                                 freq_ if which == "h" else None, sfreq)
            return lt if which == "l" else ht
        t = np.array(trans_, float).ravel() * np.ones_like(freq_)
        if np.any(t <= 0):
            raise ValueError("trans_bandwidth must be positive")
        return t

    def _length(l_tr, h_tr):
        if isinstance(filter_length, str) and filter_length.lower() == "auto":
            chk = []
            if h_tr is not None:
                chk.append(float(np.min(h_tr)))
            if l_tr is not None:
                chk.append(float(np.min(l_tr)))
            return _to_samples("%ss" % (_LENGTH_FACTOR_HAMMING / min(chk),), sfreq)
        return _to_samples(filter_length, sfreq)

    if l_freq is None:  # low-pass
        ht = _resolve_trans(h_freq, h_trans_bandwidth, "h")
        f_p, f_s = float(h_freq[0]), float(h_freq[0] + ht[0])
        if f_s > nyq:
            raise ValueError("Effective stop frequency too high")
        N = _length(None, ht)
        freq, gain = [0.0, f_p, f_s], [1, 1, 0]
        if f_s != nyq:
            freq += [nyq]
            gain += [0]
    elif h_freq is None:  # high-pass
        lt = _resolve_trans(l_freq, l_trans_bandwidth, "l")
        f_p, f_s = float(l_freq[0]), float(l_freq[0] - lt[0])
        if f_s < 0:
            raise ValueError("Filter specification invalid: Lower stop frequency negative")
        N = _length(lt, None)
        freq, gain = [f_s, f_p, nyq], [0, 1, 1]
        if f_s != 0:
            freq, gain = [0.0] + freq, [0] + gain
    elif (l_freq < h_freq).any():  # band-pass
        lt = _resolve_trans(l_freq, l_trans_bandwidth, "l")
        ht = _resolve_trans(h_freq, h_trans_bandwidth, "h")
        f_p1, f_p2 = float(l_freq[0]), float(h_freq[0])
        f_s1, f_s2 = float(l_freq[0] - lt[0]), float(h_freq[0] + ht[0])
        if f_s1 < 0:
            raise ValueError("Filter specification invalid: Lower stop frequency negative")
        if f_s2 > nyq:
            raise ValueError("Effective band-stop frequency is too high")
        N = _length(lt, ht)
        freq, gain = [f_s1, f_p1, f_p2, f_s2], [0, 1, 1, 0]
        if f_s2 != nyq:
            freq += [nyq]
            gain += [0]
        if f_s1 != 0:
            freq, gain = [0.0] + freq, [0] + gain
    else:  # band-stop (arrays allowed): create_filter(l_freq=highs, h_freq=lows)
        if len(l_freq) != len(h_freq):
            raise ValueError("l_freq and h_freq must be the same length for bandstop")
        # MNE calls _triage_filter_params(h_freq, l_freq, h_trans, l_trans, reverse=True)
        lows, highs = h_freq.copy(), l_freq.copy()
        lt = _resolve_trans(lows, h_trans_bandwidth, "l")
        ht = _resolve_trans(highs, l_trans_bandwidth, "h")
        f_p1, f_p2 = lows, highs  # pass-band edges
        f_s1, f_s2 = lows + lt, highs - ht  # stop-band edges
        if np.any(f_p1 < 0):
            raise ValueError("Filter specification invalid: Lower stop frequency negative")
        if np.any(f_p2 > nyq):
            raise ValueError("Effective band-stop frequency is too high")
        N = _length(lt, ht)
        freq = np.r_[f_p1, f_s1, f_s2, f_p2]
        gain = np.r_[np.ones_like(f_p1), np.zeros_like(f_s1), np.zeros_like(f_s2),
                     np.ones_like(f_p2)]
        order = np.argsort(freq)
        freq, gain = freq[order], gain[order]
        if freq[0] != 0:
            freq, gain = np.r_[[0.0], freq], np.r_[[1.0], gain]
        if freq[-1] != nyq:
            freq, gain = np.r_[freq, [nyq]], np.r_[gain, [1.0]]
        if np.any(np.abs(np.diff(gain, 2)) > 1):
            raise ValueError("Stop bands are not sufficiently separated.")
    freq = np.array(freq, float) / nyq
    gain = np.array(gain)
    if freq[0] != 0 or freq[-1] != 1:
        raise ValueError("freq must start at 0 and end at Nyquist")
    if N % 2 == 0:
        raise RuntimeError('filter_length must be odd if phase="zero"')
    return _firwin_design(N, freq, gain)


def _smart_pad(x: np.ndarray, n_pad: int) -> np.ndarray:
    """MNE ``_smart_pad(x, (n_pad, n_pad), "reflect_limited")`` (odd reflection)."""
    if n_pad == 0:
        return x
    z = np.zeros(max(n_pad - len(x) + 1, 0), dtype=x.dtype)
    return np.concatenate(
        [z, 2 * x[0] - x[n_pad:0:-1], x, 2 * x[-1] - x[-2 : -n_pad - 2 : -1], z]
    )


def _overlap_add_filter(x, h, n_fft=None, phase="zero", picks=None, n_jobs=1, copy=True,
                        pad="reflect_limited"):
    """Restated ``mne.filter._overlap_add_filter`` (phase="zero", reflect_limited).

    Mathematically y[n] = sum_k h[k] * x_ext[n + n_edge + (L-1)/2 - k]; the block
    FFT size affects float64 rounding only, so one full-length FFT convolution is used.
    """
    from scipy.signal import fftconvolve

    if phase != "zero" or pad != "reflect_limited":
        raise NotImplementedError
    x = np.array(x, dtype=np.float64, copy=True)
    orig_shape = x.shape
    x = np.atleast_2d(x)
    L = len(h)
    if L == 1:
        return (x * h).reshape(orig_shape)
    W = x.shape[1]
    n_edge = max(min(L, W) - 1, 0)
    shift = (L - 1) // 2 + n_edge
    out = np.empty_like(x)
    for c in range(x.shape[0]):
        x_ext = _smart_pad(x[c], n_edge)
        full = fftconvolve(x_ext, h, mode="full")
        out[c] = full[shift : shift + W]
    return out.reshape(orig_shape)


def resample(x, up=1.0, down=1.0, npad="auto", window="auto", pad="auto"):
    """Restated ``mne.filter.resample`` (method="fft", boxcar window, reflect_limited).

    PARITY UNPINNED and a no-op in every BASELINE config (ratio 1 is short-circuited
    by the reference at processing/resample.py:36-40,55-56).
    """
    from scipy.fft import rfft, irfft

    x = np.asarray(x, dtype=np.float64)
    ratio = float(up) / float(down)
    if ratio == 1.0:
        return x.copy()
    orig_shape = x.shape
    x2 = np.atleast_2d(x)
    W = x2.shape[-1]
    final_len = int(round(ratio * W))
    # npad="auto": pad to the next power of two, at least min(W // 8, 100) per side
    min_add = min(W // 8, 100) * 2
    padded = 2 ** int(np.ceil(np.log2(W + min_add)))
    npad_l = (padded - W) // 2
    npad_r = padded - W - npad_l
    orig_len = W + npad_l + npad_r
    new_len = max(int(round(ratio * orig_len)), 1)
    to_remove_l = int(round(ratio * npad_l))
    to_remove_r = new_len - final_len - to_remove_l
    out = np.empty((x2.shape[0], final_len))
    for c in range(x2.shape[0]):
        z_l = np.zeros(max(npad_l - W + 1, 0))
        z_r = np.zeros(max(npad_r - W + 1, 0))
        xe = np.concatenate([z_l, 2 * x2[c, 0] - x2[c, npad_l:0:-1], x2[c],
                             2 * x2[c, -1] - x2[c, -2 : -npad_r - 2 : -1], z_r])
        X = rfft(xe)
        # MNE _fft_resample: Nyquist bin of the shorter length is doubled (down) / halved (up)
        use_len = new_len if new_len < orig_len else orig_len
        if use_len % 2 == 0:
            X[use_len // 2] *= 2.0 if new_len < orig_len else 0.5
        y = irfft(X, n=new_len) * ratio  # irfft truncates / zero-extends the spectrum
        out[c] = y[to_remove_l : new_len - to_remove_r if to_remove_r > 0 else None][:final_len]
    return out.reshape(orig_shape[:-1] + (final_len,))

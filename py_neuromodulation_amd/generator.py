"""Window schedule of the offline stream (stream/generator.py:34-53, stream/stream.py:298,310)."""

from __future__ import annotations

import numpy as np


def window_schedule(n_samples: int, sfreq: float, sampling_rate_features_hz: float,
                    segment_length_features_ms: float):
    """-> (starts[int64], lengths[int64], time_ms[float64]).

    Float stride and segment length with ``int()`` truncation exactly like RawDataGenerator
    (e.g. 3 Hz at 1 kHz starts at 0, 333, 666, 1000, ...); iteration stops at the first window
    that would end past the data.  With a non-integer sampling rate the truncation makes the
    window length vary by one sample (1111 / 1112 at 1111.111 Hz) -- lengths are returned per
    window.  ``time`` = ceil(timestamps[-1] * 1000 + 1), timestamps = arange(start, end) / sfreq
    on the *float* start.
    """
    seg = segment_length_features_ms / 1000 * sfreq
    stride = sfreq / sampling_rate_features_hz
    # every hop at once, in the same IEEE operations as the generator's loop (k * stride, + seg, truncation): the
    # window ends grow with k, so the hops that fit are a prefix of a generous range
    n_max = max(int((n_samples - seg) / stride) + 3, 0) if stride > 0 else 0
    start = stride * np.arange(n_max, dtype=np.float64)
    end = start + seg
    fits = end.astype(np.int64) <= n_samples
    n = int(n_max if fits.all() else np.argmin(fits))
    start, end = start[:n], end[:n]
    # np.arange(start, end)[-1] without building the array: ceil((end - start) / 1) elements, the i-th is start + i
    ts_last = (start + (np.ceil(end - start) - 1)) / sfreq
    starts = start.astype(np.int64)
    return starts, end.astype(np.int64) - starts, np.ceil(ts_last * 1000 + 1)


class RawDataGenerator:
    """Drop-in for stream/generator.py: yields (timestamps, data[:, start:end])."""

    def __init__(self, data, sfreq, sampling_rate_features_hz, segment_length_features_ms):
        self.batch_counter = 0
        self.data = data
        self.sfreq = sfreq
        self.segment_length = segment_length_features_ms / 1000 * sfreq
        self.stride = sfreq / sampling_rate_features_hz

    def __iter__(self):
        return self

    def __next__(self):
        start = self.stride * self.batch_counter
        end = start + self.segment_length
        self.batch_counter += 1
        if int(end) > self.data.shape[1]:
            raise StopIteration
        return np.arange(start, end) / self.sfreq, self.data[:, int(start):int(end)]

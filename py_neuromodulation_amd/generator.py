"""Window schedule of the offline stream (stream/generator.py:34-53, stream/stream.py:298,310)."""

from __future__ import annotations

import math

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
    starts, lens, times = [], [], []
    k = 0
    while True:
        start = stride * k
        end = start + seg
        k += 1
        if int(end) > n_samples:
            break
        # np.arange(start, end)[-1] without building the array: ceil((end - start) / 1) elements, the i-th is start + i
        ts_last = (start + (math.ceil(end - start) - 1)) / sfreq
        starts.append(int(start))
        lens.append(int(end) - int(start))
        times.append(math.ceil(ts_last * 1000 + 1))
    return (np.asarray(starts, np.int64), np.asarray(lens, np.int64), np.asarray(times, np.float64))


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

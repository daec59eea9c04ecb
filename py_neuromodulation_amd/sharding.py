"""Channel sharding of the hot path over the GPUs of one node.

Every feature on the path is per channel and the only cross-channel operator (re-referencing)
is a fixed linear map of the INPUT rows, so each rank computes the features of a contiguous
block of channels from the full input: rank r applies rows [lo_r, hi_r) of the folded
(re-reference x channel-pick) matrix on its GPU.  There is NO collective on the data path; the
only communication is the final gather of the small feature table to rank 0 (control plane,
``torch.distributed.gather_object``; gloo on CPU in the tests, RCCL via "nccl" on the node).
"""

from __future__ import annotations

import numpy as np

from . import channels as chmod
from .data_processor import DataProcessor
from .generator import window_schedule
from .settings import NMSettings


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
                 world_size: int = 1, device: int | None = None, lib=None) -> None:
        self.sfreq = sfreq
        self.settings = NMSettings.load(settings)
        self.channels = chmod.load_channels(channels)
        self.line_noise = line_noise
        self.rank, self.world_size = rank, world_size
        self.device = rank if device is None else device
        self._lib = lib
        names, _, _ = chmod.channel_info(self.channels)
        self.shard = channel_shard(len(names), world_size, rank)

    def run(self, data: np.ndarray):
        """-> (local_keys, float64[n_windows, n_local], time_ms) for this rank's channels."""
        st = self.settings
        starts, lens, times = window_schedule(data.shape[1], self.sfreq, st.sampling_rate_features_hz,
                                              st.segment_length_features_ms)
        if len(set(lens.tolist())) > 1:
            raise NotImplementedError("ragged windows are not supported in sharded mode")
        dp = DataProcessor(self.sfreq, st, self.channels, line_noise=self.line_noise, verbose=False,
                           device=self.device, window=int(lens[0]) if len(lens) else None,
                           lib=self._lib, channel_subset=self.shard)
        rows = dp.process_batch(data, starts) if len(starts) else np.empty((0, len(dp.keys)))
        return list(dp.keys), rows, times


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

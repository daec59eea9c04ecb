"""Channel table handling (subset of utils/channels.py and stream/data_processor.py:141-160).

The table is a pandas DataFrame with the reference's columns
``name, rereference, used, target, type, status, new_name``.
"""

from __future__ import annotations

import numpy as np

COLUMNS = ["name", "rereference", "used", "target", "type", "status", "new_name"]


def get_default_channels_from_data(data, car_rereferencing: bool = True):
    """utils/channels.py:257-309: all channels 'ecog', good, used, common-average referenced;
    ``new_name`` is ``ch{i}_avgref`` in both modes (the reference overwrites it, :291)."""
    import pandas as pd

    n = data.shape[0]
    names = [f"ch{i}" for i in range(n)]
    return pd.DataFrame({
        "name": names,
        "rereference": ["average" if car_rereferencing else "None"] * n,
        "used": np.ones(n, dtype=int),
        "target": np.zeros(n, dtype=int),
        "type": ["ecog"] * n,
        "status": ["good"] * n,
        "new_name": [f"{c}_avgref" for c in names],
    })


def load_channels(channels):
    """DataFrame, mapping of columns, or path to a channels.csv (utils/io.py load_channels)."""
    import pandas as pd

    if isinstance(channels, pd.DataFrame):
        df = channels
    elif isinstance(channels, dict):
        df = pd.DataFrame(channels)
    else:
        df = pd.read_csv(channels)
    missing = [c for c in COLUMNS if c not in df.columns]
    if missing:
        raise ValueError(f"channels table lacks columns {missing}")
    return df.reset_index(drop=True)


def channel_info(df):
    """stream/data_processor.py:141-160 -> (ch_names_used, feature_idx, target_idx)."""
    used = df["used"].to_numpy() == 1
    good = df["status"].to_numpy() == "good"
    target = df["target"].to_numpy() == 1
    ch_names_used = df["new_name"].to_numpy()[used & good].tolist()
    feature_idx = np.flatnonzero(df["used"].to_numpy().astype(bool) & ~df["target"].to_numpy().astype(bool) & good).tolist()
    target_idx = np.flatnonzero(target).tolist()
    return ch_names_used, feature_idx, target_idx


def reref_matrix(df) -> np.ndarray | None:
    """processing/rereference.py:33-86: dense re-reference matrix over the good used channels
    (None when fewer than two channels are used)."""
    sub = df[df["used"] == 1].reset_index(drop=True)
    n = len(sub)
    if n in (0, 1):
        return None
    names = sub["name"].tolist()
    types = sub["type"].tolist()
    status = sub["status"].tolist()
    refs = sub["rereference"].tolist()
    R = np.zeros((n, n))
    types_a, good_a = np.asarray(types, dtype=object), np.asarray(status, dtype=object) == "good"
    same_type_good = {t: np.flatnonzero((types_a == t) & good_a) for t in set(types)}   # one scan per type, not per row
    for i in range(n):
        R[i, i] = 1.0
        ref = refs[i]
        if ref is None or (isinstance(ref, float) and np.isnan(ref)) or str(ref).lower() == "none" \
                or status[i] != "good":
            continue
        if ref.lower() == "average":
            grp = same_type_good[types[i]]
            idx = grp[grp != i]
        else:
            idx = []
            for rc in ref.split("&"):
                if rc not in names:
                    raise ValueError("One or more of the reference channels are not part of the "
                                     f"recording channels. First missing channel: {rc}.")
                if rc == names[i]:
                    raise ValueError(f"You cannot rereference to the same channel. Channel: {rc}.")
                idx.append(names.index(rc))
        R[i, idx] = -1.0 / len(idx)
    good = [i for i in range(n) if status[i] == "good"]
    return R[np.ix_(good, good)]


def reref_structure(full: np.ndarray, max_taps: int = 4):
    """Decompose the rows of a (folded) re-reference matrix into explicit taps plus a multiple of ONE group sum:

        y_r = sum_k coef[r][k] * x[idx[r][k]]  +  b[r] * sum_{j in groups[g[r]]} x_j

    (the host-side twin of find_reref_structure in csrc/nmx_engine.inc).  "average" rows of
    processing/rereference.py:61-63 -- 1 on the channel itself, -1/(n-1) on every other good channel of its
    type -- become one tap (1 + 1/(n-1)) plus b = -1/(n-1) times the sum over the WHOLE type group, shared by
    all rows of the group; bipolar rows (:65-79) are taps only.
    Returns (taps: list of [(col, coef), ...], g: int array (-1 = none), b: float array, groups: list of
    sorted column arrays), or None when some row has no such structure."""
    full = np.asarray(full, np.float64)
    taps, g, b, groups, index = [], [], [], [], {}
    for row in full:
        nz = np.flatnonzero(row)
        if len(nz) <= max_taps:
            taps.append([(int(j), float(row[j])) for j in nz])
            g.append(-1)
            b.append(0.0)
            continue
        vals, counts = np.unique(row[nz], return_counts=True)
        o = float(vals[np.argmax(counts)])
        rest = [int(j) for j in nz if row[j] != o]
        if len(rest) > max_taps:
            return None
        key = tuple(int(j) for j in nz)
        if key not in index:
            index[key] = len(groups)
            groups.append(np.asarray(nz, dtype=np.int64))
        taps.append([(j, float(row[j]) - o) for j in rest])
        g.append(index[key])
        b.append(o)
    return taps, np.asarray(g, dtype=np.int64), np.asarray(b, dtype=np.float64), groups


def split_hi_lo(v: np.ndarray) -> np.ndarray:
    """float64 [T] -> float32 [2, T]: hi = float32(v), lo = float32(v - hi); hi + lo carries ~48 bits of v.  The group
    sums of a channel shard travel this way (sharding.py): a float32 sum of 256 channels with +-500 offsets would add a
    rounding step the single-device kernel does not have."""
    fmax = float(np.finfo(np.float32).max)
    # (a sum beyond float32's range -- several members at +-inf, which nan_to_num turns into +-3.4e38 each -- would give
    # hi = +inf, lo = -inf: the device cleans those to +-FLT_MAX and the pair cancels to 0.  Both halves saturate instead:
    # two members on the rail travel exactly, as (FLT_MAX, FLT_MAX) -- the kernels add the halves in float64 --, more
    # than two as a huge value of the right sign.)
    v = np.asarray(v, np.float64)
    hi = np.clip(v, -fmax, fmax).astype(np.float32)
    lo = np.clip(v - hi.astype(np.float64), -fmax, fmax).astype(np.float32)
    return np.stack([hi, lo])

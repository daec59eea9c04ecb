"""Output files of a stream run, same names and layouts as the reference:

  {out_dir}/{name}/{name}-{i}.msgpack   list of {key: float} dicts, one per hop, one file per save
                                        interval (utils/file_writer.py:26-118)
  {out_dir}/{name}/{name}_FEATURES.csv  all hops (utils/file_writer.py:92-105, utils/io.py:246-262)
  {out_dir}/{name}/{name}_SIDECAR.json  {original_fs, final_fs, sfreq, sess_right, ...}
                                        (stream/data_processor.py:313-337, utils/io.py:265-293)
  {out_dir}/{name}/{name}_SETTINGS.yaml, {name}_channels.csv   (stream/stream.py:426-441)

The batch engine produces all hops at once, so ``insert_rows`` takes the feature matrix and the
writer cuts it into the same per-interval files a hop-by-hop run would have produced.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np


class MsgPackFileWriter:
    def __init__(self, name: str = "sub", out_dir="") -> None:
        self.out_dir = (Path.cwd() if not out_dir else Path(out_dir)) / name
        self.out_dir.mkdir(parents=True, exist_ok=True)
        self.idx = 0
        self.name = name
        self.csv_path = self.out_dir / f"{name}_FEATURES.csv"
        self.data_l: list[dict] = []

    # -- hop-by-hop interface (the reference's) ------------------------------------------------
    def insert_data(self, feature_dict: dict) -> None:
        for key, value in feature_dict.items():
            feature_dict[key] = float(value) if value is not None else 0
        self.data_l.append(feature_dict)

    def save(self) -> None:
        import msgpack

        if not self.data_l:
            return
        with open(self.out_dir / f"{self.name}-{self.idx}.msgpack", "wb") as f:
            msgpack.pack(self.data_l, f)
        self.idx += 1
        self.data_l = []

    # -- batch interface -----------------------------------------------------------------------
    def insert_rows(self, keys, rows: np.ndarray, save_interval: int = 10) -> None:
        """rows[n_hops, len(keys)]: one msgpack file per ``save_interval`` hops, like a run that
        calls insert_data per hop and save() every ``save_interval`` hops (stream/stream.py:326-331)."""
        keys = [str(k) for k in keys]
        for i, row in enumerate(np.asarray(rows, dtype=np.float64).tolist()):
            self.data_l.append(dict(zip(keys, row)))
            if (i + 1) % save_interval == 0:
                self.save()
        self.save()

    def load_all(self):
        import msgpack
        import pandas as pd

        data_l = []
        for i in range(self.idx):
            with open(self.out_dir / f"{self.name}-{i}.msgpack", "rb") as f:
                data_l.extend(msgpack.unpack(f))
        if not data_l:
            raise ValueError("No data to load")
        return pd.DataFrame(data_l)

    def save_as_csv(self, save_all_combined: bool = False) -> None:
        import msgpack
        import pandas as pd

        if save_all_combined:
            self.load_all().to_csv(self.csv_path, index=False)
            return
        if self.data_l:
            pd.DataFrame(self.data_l[-1:]).to_csv(self.csv_path, index=False)
        else:
            with open(self.out_dir / f"{self.name}-0.msgpack", "rb") as f:
                pd.DataFrame(msgpack.unpack(f)).to_csv(self.csv_path, index=False)

    def delete_ind_files(self) -> None:
        for file in self.out_dir.glob(f"{self.name}-*.msgpack"):
            file.unlink()


def _json_default(obj):
    import pandas as pd

    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, pd.DataFrame):
        return obj.to_numpy().tolist()
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    raise TypeError("Not serializable")


def save_sidecar(sidecar: dict, out_dir="", prefix: str = "") -> Path:
    out_dir = Path.cwd() if not out_dir else Path(out_dir)
    path = out_dir / prefix / f"{prefix}_SIDECAR.json"
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        json.dump(sidecar, f, default=_json_default, indent=4, separators=(",", ": "))
    return path


def save_features(df, out_dir="", prefix: str = "") -> Path:
    out_dir = Path.cwd() if not out_dir else Path(out_dir)
    path = out_dir / (f"{prefix}_FEATURES.csv" if prefix else "_FEATURES.csv")
    path.parent.mkdir(parents=True, exist_ok=True)
    df.to_csv(path, index=False)
    return path


def channels_csv_text(channels) -> str:
    # utils/io.py:234-253 writes through pyarrow.csv: strings (and the header) quoted, numbers bare
    import csv

    return channels.to_csv(None, index=False, quoting=csv.QUOTE_NONNUMERIC, lineterminator="\n")


def save_channels(channels, out_dir="", prefix: str = "", text: str | None = None) -> Path:
    out_dir = Path.cwd() if not out_dir else Path(out_dir)
    path = out_dir / prefix / ("channels.csv" if not prefix else prefix + "_channels.csv")
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w", newline="") as f:
        f.write(text if text is not None else channels_csv_text(channels))
    return path

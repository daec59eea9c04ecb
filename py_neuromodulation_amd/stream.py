"""Offline stream driver: drop-in for ``nm.Stream`` (stream/stream.py:22-345).

``Stream(...).run(data) -> pd.DataFrame`` with the reference's columns (features in reference
order, then ``time`` (:310), then target channels (:145-170)).  Where the reference loops over
hops on one CPU thread, the whole recording is put on the GPU once and all hops are computed in
one batch (windows are strided views of the resident array); only the sequential, tiny
post-processing (normaliser) runs per hop on the host.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np

from . import channels as chmod
from .data_processor import DataProcessor
from .generator import window_schedule
from .settings import NMSettings

_FREQ_FEATURES = ["bandpass_filter", "stft", "fft", "welch", "bursts", "coherence", "nolds", "bispectrum"]


class Stream:
    def __init__(self, sfreq: float, channels=None, data=None, settings=None,
                 line_noise: float | None = 50, sampling_rate_features_hz: float | None = None,
                 path_grids=None, coord_names=None, coord_list=None, verbose: bool = False,
                 device: int = 0, lib=None, resample_features_at_new_rate: bool = False,
                 devices=None) -> None:
        """``devices=[0, 1, ...]``: the channels are sharded in contiguous blocks over these GPUs inside this one
        process (one plan and one host thread per device, sharding.MultiDeviceProcessor); ``device`` otherwise."""
        self.settings = NMSettings.load(settings)
        if channels is None and data is not None:
            channels = chmod.get_default_channels_from_data(data)
        if channels is None and data is None:
            raise ValueError("Either `channels` or `data` must be passed to `Stream`.")
        self.channels = chmod.load_channels(channels)
        if not np.any((self.channels["used"].to_numpy() == 1) & (self.channels["target"].to_numpy() == 0)):   # (DataFrame.query: 2 ms)
            raise ValueError("No channels selected for analysis that have column 'used' = 1 and "
                             "'target' = 0. Please check your channels")
        if any(f in _FREQ_FEATURES for f in self.settings.features.get_enabled()):
            assert all(fb[1] < sfreq / 2 for fb in self.settings.frequency_ranges_hz.values()), (
                "If a feature that uses frequency ranges is selected, the frequency band ranges "
                f"need to be smaller than the nyquist frequency.\nGot sfreq = {sfreq} and fband "
                f"ranges:\n {self.settings.frequency_ranges_hz}")
        if sampling_rate_features_hz is not None:
            self.settings.sampling_rate_features_hz = sampling_rate_features_hz
        self.sfreq = sfreq
        self.line_noise = line_noise
        self.verbose = verbose
        self.device = device
        self.devices = [int(d) for d in devices] if devices is not None else None
        self._resample_new_rate = resample_features_at_new_rate
        self._lib = lib  # None = the product library (libnmx.so); tests may inject a binding
        self.data = data
        self.sess_right = None
        self.is_running = False
        # fail early like the reference (it builds a DataProcessor in __init__, :130)
        self.data_processor = self._make_processor(None)

    def _settings_token(self):
        """Changes whenever the settings / channel table the processor was built from change."""
        import json

        return json.dumps(self.settings.to_dict(), sort_keys=True, default=str) + self.channels.to_json()

    def _processor_token(self, settings_token=None):
        """Everything a DataProcessor is built from: the settings / channel table AND the stream attributes the
        reference reads afresh on every run (stream/stream.py:233-242 builds a new DataProcessor per run)."""
        from . import user_features

        return (settings_token if settings_token is not None else self._settings_token(), repr(self.line_noise),
                repr(self.sfreq), bool(self._resample_new_rate),
                int(self.device), tuple(self.devices or ()), id(self._lib),
                tuple((k, id(v)) for k, v in user_features.items()))

    def _make_processor(self, window):
        if self.devices is not None and len(self.devices) > 1:
            from .sharding import MultiDeviceProcessor

            dp = MultiDeviceProcessor(self.sfreq, self.settings, self.channels, line_noise=self.line_noise,
                                      devices=self.devices, window=window, lib=self._lib, verbose=self.verbose,
                                      resample_features_at_new_rate=self._resample_new_rate)
            dp.settings_token = self._processor_token()
            return dp
        dp = DataProcessor(sfreq=self.sfreq, settings=self.settings, channels=self.channels,
                           line_noise=self.line_noise, verbose=self.verbose,
                           device=self.devices[0] if self.devices else self.device,
                           window=window, lib=self._lib,
                           resample_features_at_new_rate=self._resample_new_rate)
        dp.settings_token = self._processor_token()
        return dp

    def _handle_data(self, data) -> np.ndarray:
        names_expected = self.channels["name"].to_list()
        if isinstance(data, np.ndarray):
            if len(names_expected) != data.shape[0]:
                raise ValueError("If data is passed as an array, the first dimension must match the "
                                 f"number of channel names in `channels`.\n Number of data channels "
                                 f"(data.shape[0]): {data.shape[0]}\n Length of channels[\"name\"]: "
                                 f"{len(names_expected)}.")
            return data
        names_data = data.columns.to_list()
        if not (len(names_expected) == len(names_data) and sorted(names_expected) == sorted(names_data)):
            raise ValueError("If data is passed as a DataFrame, the column names must match the channel "
                             f"names in `channels`.\nInput dataframe column names: {names_data}\n"
                             f"Expected (from channels[\"name\"]): : {names_expected}.")
        return data.to_numpy().transpose()

    def run(self, data=None, out_dir="", experiment_name: str = "sub", is_stream_lsl: bool = False,
            stream_lsl_name: str | None = None, save_csv: bool = True, save_interval: int = 10,
            return_df: bool = True, simulate_real_time: bool = False, decoder=None, backend_interface=None,
            delete_ind_batch_files_after_stream: bool = True, save_msgpack: bool | None = None):
        """Compute every hop of ``data`` and return the feature DataFrame (signature and defaults of
        stream/stream.py:198-212).  The reference writes a ``{name}-{i}.msgpack`` file every ``save_interval``
        hops and deletes them after the run unless ``delete_ind_batch_files_after_stream=False``; the batch
        driver only writes them when they are to be kept (``save_msgpack`` forces either way)."""
        if is_stream_lsl or stream_lsl_name is not None:
            raise NotImplementedError("LSL acquisition is outside the accelerated hot path (SURVEY.md section 2)")
        if decoder is not None or backend_interface is not None or simulate_real_time:
            raise NotImplementedError("real-time decoding / GUI back-end / real-time simulation are outside the "
                                      "accelerated hot path (SURVEY.md section 2)")
        if save_msgpack is None:
            save_msgpack = not delete_ind_batch_files_after_stream
        # what the reference's run leaves on the object (stream/stream.py:213-219) -- its examples read `stream.out_dir` and
        # `stream.experiment_name` afterwards (examples/plot_0_first_demo.py: nm.FeatureReader(feature_dir=stream.out_dir, ...))
        self.is_stream_lsl, self.stream_lsl_name = is_stream_lsl, stream_lsl_name
        self.save_csv, self.save_interval, self.return_df = save_csv, save_interval, return_df
        self.out_dir = Path.cwd() if not out_dir else Path(out_dir)
        self.experiment_name = experiment_name
        import pandas as pd

        if data is not None:
            data = self._handle_data(data)
        elif self.data is not None:
            data = self._handle_data(self.data)
        else:
            raise ValueError("No data passed to run function.")
        self.is_running = True
        st = self.settings
        settings_token = self._settings_token()   # (serialises settings and channel table: once per run)
        starts, lens, times = window_schedule(data.shape[1], self.sfreq, st.sampling_rate_features_hz,
                                              st.segment_length_features_ms)
        rows = np.empty((len(starts), 0))
        keys: list[str] = []
        tgt_rows = np.flatnonzero(self.channels["target"].to_numpy() == 1)   # (rows of `data`, in table order)
        tgt_names = [str(v) for v in self.channels["name"].to_numpy()[tgt_rows]]
        n_targets = len(tgt_rows)
        # the side-car files do not depend on the features (stream/stream.py:338 writes them after the loop): written on a
        # thread of their own while the device works
        import threading

        after_err: list = []

        def write_after():
            try:
                self._save_after_stream(out_dir, experiment_name, settings_token)
            except BaseException as e:   # noqa: BLE001 -- re-raised by run()
                after_err.append(e)

        after = None
        if len(starts):
            groups = [int(g) for g in np.unique(lens)]
            # a FRESH processing state per run, like the reference's new DataProcessor (:233-242), one
            # processor per window length.  The processor built by __init__ (or by the previous run) is
            # reused with its state reset when it fits -- same results, no second plan build.
            procs = {}
            for w in groups:
                dp = self.data_processor
                if (len(groups) == 1 and dp is not None and dp.engine.W_in == w
                        and dp.settings_token == self._processor_token(settings_token)):
                    dp.reset()
                    procs[w] = dp
                else:
                    procs[w] = self._make_processor(w)
            self.data_processor = dp0 = procs[groups[0]]
            if len(groups) == 1:
                side = self._side_files(out_dir, experiment_name)
                had = [f.exists() for f in side]
                after = threading.Thread(target=write_after, daemon=True)
                after.start()
                try:
                    rows = dp0.process_batch(data, starts, spare_cols=1 + n_targets)   # ("time" and the targets behind the features)
                except BaseException as e:
                    # the reference writes these files after its loop (stream/stream.py:338): a run that fails leaves none
                    after.join()
                    for f, was in zip(side, had):
                        if not was:
                            f.unlink(missing_ok=True)
                    if after_err:
                        raise e from after_err[0]
                    raise
            else:
                # ragged windows (float sampling rate): the normaliser is sequential over ALL hops, so it
                # cannot run inside the per-length engines: detached, it normalises the merged rows afterwards.
                # Hops in ORDER, as consecutive runs of one window length: the burst history (features/bursts.py:149-173)
                # and the Kalman filters of the band powers (bandpower.py:147-163) carry over from hop to hop whatever
                # the window length, so the state travels from processor to processor where the length changes
                # (DataProcessor.ragged_* / MultiDeviceProcessor.ragged_*: one state blob per device there)
                for p in procs.values():
                    p.ragged_prepare()
                cuts = [0] + [i for i in range(1, len(lens)) if lens[i] != lens[i - 1]] + [len(lens)]
                state, runs = None, []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    p = procs[int(lens[a])]
                    if state is not None:
                        p.ragged_set_state(state)
                    runs.append(p.ragged_run(data, starts[a:b]))
                    state = p.ragged_state()
                rows = dp0.ragged_finish(runs)
            keys = list(dp0.keys)   # after the first hop: user-feature keys are known once calc_feature has run
        last = starts + lens - 1   # the targets' column: the last sample of every window (stream/stream.py:319-329)
        if len(keys) and rows.shape[1] == len(keys) + 1 + n_targets:   # the table came with room for them: one block, no inserts
            rows[:, len(keys)] = times
            for k, idx in enumerate(tgt_rows):
                rows[:, len(keys) + 1 + k] = np.asarray(data[idx])[last]
            df = pd.DataFrame(rows, columns=self._column_index(keys + ["time"] + tgt_names))
        else:
            df = pd.DataFrame(rows, columns=self._column_index(keys))
            df["time"] = times
            for idx, name in zip(tgt_rows, tgt_names):
                df[name] = np.asarray(data[idx], dtype=np.float64)[last]
        self.is_running = False
        self.batch_count = len(df)   # (stream/stream.py:229,317: hops processed)
        # ---- output files, names and layouts of the reference (stream/stream.py:229,319-343,426-453)
        writer = None
        if save_msgpack:
            from .file_writer import MsgPackFileWriter

            writer = MsgPackFileWriter(name=experiment_name, out_dir=out_dir)
            writer.insert_rows(df.columns, df.to_numpy(dtype=np.float64), save_interval=save_interval)
        if save_csv:
            out = (Path.cwd() if not out_dir else Path(out_dir)) / experiment_name
            out.mkdir(parents=True, exist_ok=True)
            df.to_csv(out / f"{experiment_name}_FEATURES.csv", index=False)
        if after is not None:
            after.join()
            if after_err:
                raise after_err[0]
        else:
            self._save_after_stream(out_dir, experiment_name, settings_token)   # always, like the reference (stream/stream.py:338)
        if writer is not None and delete_ind_batch_files_after_stream:
            writer.delete_ind_files()
        return df if return_df else {}

    _col_indexes: list = []   # (names, Index) of the last few result tables, shared by every Stream of the process
    _after_texts: list = []   # (settings token, SETTINGS.yaml text, channels.csv text) of the last few runs

    def _column_index(self, names: list):
        """The column Index of the result table, built once per set of names: hashing ~8000 strings into a new Index is
        0.5 ms of every run (an Index is immutable: the frames of consecutive runs -- of this Stream or of a fresh one with
        the same settings -- share it)."""
        import pandas as pd

        cache = Stream._col_indexes
        for entry in cache:
            if entry[0] == names:   # (mostly the same string objects: compared by identity first)
                return entry[1]
        entry = (list(names), pd.Index(names))
        cache.insert(0, entry)
        del cache[4:]
        return entry[1]

    @staticmethod
    def _side_files(out_dir="", experiment_name: str = "sub"):
        base = (Path.cwd() if not out_dir else Path(out_dir)) / experiment_name
        return [base / f"{experiment_name}_SIDECAR.json", base / f"{experiment_name}_SETTINGS.yaml",
                base / f"{experiment_name}_channels.csv"]

    # -- stream/stream.py:426-453, stream/data_processor.py:313-337 -----------------------------
    def _save_after_stream(self, out_dir="", experiment_name: str = "sub", settings_token=None) -> None:
        from . import file_writer as fw

        # stream/data_processor.py:313-337: original_fs = the rate passed in, final_fs = sfreq // 1
        sidecar = {"original_fs": self.sfreq, "final_fs": self.data_processor.sfreq_raw,
                   "sfreq": float(self.settings.sampling_rate_features_hz), "sess_right": self.sess_right}
        fw.save_sidecar(sidecar, out_dir, experiment_name)
        # the serialised settings / channel table of the previous run are re-used while both are unchanged
        # (process-wide: a fresh Stream with the settings of the one before -- the reference builds one per run -- does not
        # serialise them again; the YAML dump is 3 - 5 ms of pure Python on this thread, holding the GIL the main thread
        # needs to start its pipeline)
        token = settings_token if settings_token is not None else self._settings_token()
        cache = next((c for c in Stream._after_texts if c[0] == token), None)
        if cache is None:
            cache = (token, self.settings.to_yaml_text(), fw.channels_csv_text(self.channels))
            Stream._after_texts.insert(0, cache)
            del Stream._after_texts[4:]
        self.settings.save(out_dir or Path.cwd(), experiment_name, text=cache[1])
        fw.save_channels(self.channels, out_dir, experiment_name, text=cache[2])

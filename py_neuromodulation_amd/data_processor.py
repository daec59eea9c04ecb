"""Per-window orchestrator: drop-in for ``DataProcessor`` (stream/data_processor.py:19-311)
plus a batch-of-windows entry point.

process(data[C_all, W]) reproduces, in order: NaN mask over all incoming rows (:253),
nan_to_num + channel pick (:255), pre-processors in PREPROCESSOR_DICT order filtered by
membership in settings.preprocessing (processing/data_preprocessor.py:9-15,45-51), all enabled
features in FeatureSelector order (features/feature_processor.py:45-84), feature normalisation
of the non-"psd" keys (:263-290), and the NaN policy: every key that CONTAINS the new_name of
a NaN channel becomes NaN (:297-306, substring match, reproduced as is).

Everything up to the features is ONE launch sequence on the GPU: the channel pick and the
re-reference matrix are folded into one [C, C_all] matrix applied on the device, notch runs per
window on the device, features read the result from HBM/LDS.

User-registered features (``add_custom_feature``, features/feature_processor.py:52-53,90-108) are
instantiated after the built-in ones and called on the host, hop by hop, with the pre-processed window
the device features read (``nmx_process_batch_tap``; the float64 window itself when nothing is enabled in
``settings.preprocessing`` and every channel is picked); their keys follow the built-in columns in
registration order and go through the same normaliser / NaN policy.
"""

from __future__ import annotations

from time import time

import numpy as np

from . import channels as chmod
from . import fir_design
from .engine import HotPathEngine
from .processing import DeviceFeatureNormalizer, FeatureNormalizer
from .settings import NMSettings

PREPROCESSOR_ORDER = ["preprocessing_filter", "notch_filter", "raw_resampling", "re_referencing",
                      "raw_normalization"]


class _LazyNanCols:
    def __init__(self, keys, ch_names) -> None:
        self._keys, self._ch, self._cache = keys, ch_names, {}

    def __getitem__(self, ci: int) -> np.ndarray:
        cols = self._cache.get(ci)
        if cols is None:
            ch = self._ch[ci]
            cols = self._cache[ci] = np.array([i for i, k in enumerate(self._keys) if ch in k], dtype=int)
        return cols


class UserColumns:
    """The user-registered features of one stream (features/feature_processor.py:52-53): instances built after the
    built-in features with the same ``(settings, ch_names, sfreq)``, called hop by hop on the host with the
    pre-processed window (``estimate_features``, :80-82).  ``keys`` is the orchestrator's column list: the first call
    appends the plugin keys to it (a key that repeats an earlier one overwrites that column, ``dict.update``).
    The plugin columns go through their own device normaliser (per-column statistics: splitting the table changes
    nothing) in float32 like the built-in columns."""

    def __init__(self, settings, ch_names, sfreq, keys: list, device: int = 0, lib=None) -> None:
        self.settings, self.ch_names, self.sfreq, self.keys = settings, list(ch_names), sfreq, keys
        self.device, self.lib = int(device), lib
        self.user_keys: list[str] | None = None
        self.cols: np.ndarray | None = None
        self.norm = None
        self.n_builtin = len(keys)
        self._instantiate()

    def _instantiate(self) -> None:
        from . import user_features as registered

        self.features = {name: cls(self.settings, self.ch_names, self.sfreq) for name, cls in registered.items()}

    def reset(self) -> None:
        """Fresh instances and history, like the reference's new FeatureProcessors per DataProcessor."""
        self._instantiate()
        if self.norm is not None:
            self.norm.reset()

    def rows(self, windows) -> np.ndarray:
        """windows: iterable of float64 [C, W] pre-processed windows in hop order -> float64 [n, n_user]."""
        rows = []
        for w in windows:
            d: dict = {}
            for f in self.features.values():
                d.update(f.calc_feature(w))
            if self.user_keys is None:
                self.user_keys = list(d)
                col = {k: i for i, k in enumerate(self.keys)}
                n0 = len(self.keys)
                new = [k for k in self.user_keys if k not in col]
                self.keys.extend(new)
                col.update({k: n0 + i for i, k in enumerate(new)})
                self.cols = np.array([col[k] for k in self.user_keys], dtype=np.int64)
                st = self.settings
                if st.postprocessing.feature_normalization:
                    mask = None
                    if not st.feature_normalization_settings.normalize_psd:
                        mask = np.array(["psd" not in k for k in self.user_keys], dtype=np.uint8)
                    self.norm = DeviceFeatureNormalizer(st, len(self.user_keys), colmask=mask, device=self.device,
                                                        lib=self.lib)
            if list(d) != self.user_keys:
                raise ValueError("a user feature changed its keys between hops: "
                                 f"{sorted(set(d) ^ set(self.user_keys))[:4]}")
            rows.append(np.fromiter(d.values(), dtype=np.float64, count=len(d)))
        out = np.stack(rows) if rows else np.empty((0, len(self.user_keys or [])))
        if self.norm is not None and len(out):
            out = self.norm.process_batch(out).astype(np.float64)
        return out

    def merge(self, builtin: np.ndarray, user: np.ndarray) -> np.ndarray:
        out = np.empty((builtin.shape[0], len(self.keys)), np.float64)
        out[:, :builtin.shape[1]] = builtin
        if user.shape[0]:
            out[:, self.cols] = user     # after the built-in values: dict.update overwrites a repeated key
        return out


class DataProcessor:
    def __init__(self, sfreq: float, settings, channels, coord_names=None, coord_list=None,
                 line_noise: float | None = None, path_grids=None, verbose: bool = True,
                 device: int = 0, window: int | None = None, lib=None,
                 channel_subset=None, dry_run: bool = False,
                 resample_features_at_new_rate: bool = False, local_inputs: bool = False,
                 staging_slot: int = 0) -> None:
        self.settings = NMSettings.load(settings)
        self.channels = chmod.load_channels(channels)
        # (what a twin for another window length is built from: `process` under the reference's own loop, below)
        self._ctor = dict(sfreq=sfreq, line_noise=line_noise, verbose=verbose, device=device, lib=lib,
                          resample_features_at_new_rate=resample_features_at_new_rate, staging_slot=staging_slot)
        self._twin_ok = channel_subset is None and not dry_run and not local_inputs
        self._by_len, self._active = None, None
        self.sfreq_features = self.settings.sampling_rate_features_hz
        self._sfreq_raw_orig = sfreq
        self.sfreq_raw = sfreq // 1
        self.line_noise = line_noise
        self.verbose = verbose
        st = self.settings
        if st.postprocessing.project_cortex or st.postprocessing.project_subcortex:
            raise NotImplementedError("grid projection is outside the accelerated hot path")
        self.ch_names_used, self.feature_idx, self.target_idx = chmod.channel_info(self.channels)
        n_all = len(self.channels)

        notch_taps = None
        R = None
        resample_to = None
        pre_taps = None
        raw_norm = None
        for name in st.preprocessing:
            if name not in PREPROCESSOR_ORDER:
                raise ValueError(f"Invalid preprocessing method '{name}'. Must be one of {PREPROCESSOR_ORDER}")
        for name in PREPROCESSOR_ORDER:
            if name not in st.preprocessing:
                continue
            if name == "notch_filter":
                if line_noise is None:
                    raise ValueError("Either line_noise or freqs must be defined if notch_filter is activated.")
                notch_taps = fir_design.notch_bank(self.sfreq_raw, line_noise)
            elif name == "preprocessing_filter":
                pre_taps = fir_design.preprocessing_filter_bank(st.preprocessing_filter, self.sfreq_raw)
            elif name == "raw_resampling":
                new_rate = float(st.raw_resampling_settings.resample_freq_hz)
                if float(new_rate / self.sfreq_raw) != 1.0:
                    # The reference resamples every window but keeps building notch AND features with the
                    # RAW rate (stream/data_processor.py:55,68,80): every frequency axis of the features is
                    # off by the ratio.  Reproduced as is (default); resample_features_at_new_rate=True
                    # selects the consistent pipeline (features designed for the rate they see).
                    resample_to = new_rate
                    if not resample_features_at_new_rate:
                        from . import logger

                        logger.info("raw_resampling %g -> %g Hz: features are designed with the raw rate like "
                                    "the reference (pass resample_features_at_new_rate=True for the new rate)",
                                    self.sfreq_raw, new_rate)
            elif name == "re_referencing":
                R = chmod.reref_matrix(self.channels)
            elif name == "raw_normalization":
                rs = st.raw_normalization_settings
                rate = resample_to if (resample_to is not None and resample_features_at_new_rate) else self.sfreq_raw
                # normalization.py:52-58: add_samples = int(sfreq / feat_hz), N = int(time_s * sfreq)
                raw_norm = (rs.normalization_method, rs.clip, int(rs.normalization_time_s * rate),
                            int(rate / st.sampling_rate_features_hz))
            else:
                raise NotImplementedError(f"{name} is outside the accelerated hot path (SURVEY 8f)")
        C = len(self.feature_idx)
        if R is not None and R.shape[0] != C:
            raise ValueError(f"re-reference matrix is {R.shape} but {C} channels are picked for features")
        # fold data[feature_idx] (and R) into one [C, C_all] matrix applied on the device
        need_matrix = R is not None or self.feature_idx != list(range(n_all))
        full = None
        if need_matrix:
            S = np.zeros((C, n_all))
            S[np.arange(C), self.feature_idx] = 1.0
            full = (R @ S) if R is not None else S
        self.ref_matrix = R
        # channel sharding over GPUs: this processor computes the features of a subset of the
        # picked channels but its re-reference rows still read ALL input rows
        self.all_ch_names_used = list(self.ch_names_used)
        names = self.ch_names_used
        self.local_rows = None      # local_inputs: the input rows this processor expects, then hi / lo rows of the group sums
        self.local_groups = []      # local_inputs: member rows (global) of every group sum it expects
        if channel_subset is not None:
            subset = list(channel_subset)
            if full is None:
                full = np.eye(n_all)
            full = full[subset]
            names = [self.ch_names_used[i] for i in subset]
            if local_inputs:
                # Channel shard WITHOUT replicating the recording: this rank is handed only the input rows its
                # own output rows tap, plus one row per type group holding sum_{j in group} x_j (partial sums
                # of the owners, all-reduced by the caller -- sharding.ShardedStream).  Its matrix then has a
                # handful of non-zeros per row: taps on local rows + the coefficient of the group-sum row.
                st_ = chmod.reref_structure(full)
                if st_ is None:
                    raise NotImplementedError("local_inputs needs re-reference rows made of a few named channels "
                                              "and / or one group average (processing/rereference.py:52-86)")
                taps, gi, gb, groups = st_
                rows = sorted({j for t in taps for j, _ in t})
                pos = {j: i for i, j in enumerate(rows)}
                used_groups = sorted({int(k) for k in gi if k >= 0})
                # a group sum arrives as TWO float32 rows, hi + lo of the float64 sum (channels.split_hi_lo): the
                # device accumulates coefficient x row in float64, so the re-referenced sample is rounded to float32
                # once, as in the single-device kernel that forms the sum itself
                gpos = {k: len(rows) + 2 * i for i, k in enumerate(used_groups)}
                loc = np.zeros((len(subset), len(rows) + 2 * len(used_groups)))
                for r, t in enumerate(taps):
                    for j, c in t:
                        loc[r, pos[j]] += c
                    if gi[r] >= 0:
                        loc[r, gpos[int(gi[r])]] = gb[r]
                        loc[r, gpos[int(gi[r])] + 1] = gb[r]
                full = loc if not (loc.shape[0] == loc.shape[1] and np.array_equal(loc, np.eye(len(loc)))) else None
                self.local_rows = rows
                self.local_groups = [groups[k] for k in used_groups]
        if resample_to is None:
            self.engine = HotPathEngine(st, names, self.sfreq_raw, ref_matrix=full, notch_taps=notch_taps,
                                        device=device, window=window, lib=lib, dry_run=dry_run,
                                        pre_taps=pre_taps, raw_norm=raw_norm, staging_slot=staging_slot)
        elif resample_features_at_new_rate:   # `window` counts RAW samples (the generator cuts raw data)
            self.engine = HotPathEngine(st, names, resample_to, ref_matrix=full, notch_taps=notch_taps,
                                        device=device, lib=lib, dry_run=dry_run,
                                        resample_from=self.sfreq_raw, raw_window=window, pre_taps=pre_taps,
                                        raw_norm=raw_norm, staging_slot=staging_slot)
            self.sfreq_raw = resample_to
        else:   # the reference: windows resampled, everything designed with the raw rate
            self.engine = HotPathEngine(st, names, self.sfreq_raw, ref_matrix=full, notch_taps=notch_taps,
                                        device=device, lib=lib, dry_run=dry_run,
                                        resample_from=self.sfreq_raw, resample_to=resample_to,
                                        raw_window=window, pre_taps=pre_taps, raw_norm=raw_norm, staging_slot=staging_slot)
        self.keys = list(self.engine.keys)
        # user features: instantiated after the built-ins with the same arguments (feature_processor.py:45-53); a
        # channel shard leaves them to its coordinator (they see ALL channels: sharding.MultiDeviceProcessor)
        from . import user_features as _registered

        self._user = None
        if _registered and not dry_run and channel_subset is None:
            self._user = UserColumns(st, names, self.sfreq_raw, self.keys, device=device, lib=lib)
        self._user_chunk = 64                          # hops per tapped batch (bounds the [n, C, W] hand-back)
        self.feature_normalizer = None
        self.non_psd_indices = None
        self.device_normalizer = None
        self._norm_in_engine = False
        if st.postprocessing.feature_normalization:
            fs = st.feature_normalization_settings
            if not fs.normalize_psd:
                self.non_psd_indices = np.array([i for i, k in enumerate(self.keys) if "psd" not in k], dtype=int)
            if fs.normalization_method in DeviceFeatureNormalizer.METHODS and not dry_run:
                # "mean" / "zscore" (default): one HIP scan per batch of hops; the column mask
                # carries the "psd" exclusion (stream/data_processor.py:263-290)
                mask = None
                if self.non_psd_indices is not None:
                    mask = np.zeros(len(self.keys), dtype=np.uint8)
                    mask[self.non_psd_indices] = 1
                self.device_normalizer = DeviceFeatureNormalizer(st, len(self.keys), colmask=mask,
                                                                 device=device, lib=lib)
                # inside the engine's launch sequence: rows come back normalised (no second round trip)
                self.engine.attach_normalizer(self.device_normalizer)
                self._norm_in_engine = True
            elif not dry_run:   # an unknown method name (every method of normalization.py:57-70 runs on the device): raises, naming it
                self.feature_normalizer = FeatureNormalizer(st)
        # NaN policy: columns whose key contains the channel's new_name (substring, as the reference);
        # built on first use per channel (256 channels x 8 000 keys of substring tests cost 50 ms up front,
        # and a NaN channel is the exception)
        self._nan_cols = _LazyNanCols(self.keys, self.ch_names_used)
        self.cnt_samples = 0
        self.settings_token = None

    # -- stream/data_processor.py:313-351: what the reference's Stream calls after its loop ---------------------------------
    def save_sidecar(self, out_dir, prefix: str = "", additional_args: dict | None = None) -> None:
        from . import file_writer as fw

        sidecar = {"original_fs": self._sfreq_raw_orig, "final_fs": self.sfreq_raw, "sfreq": self.sfreq_features}
        if additional_args is not None:
            sidecar = sidecar | additional_args
        fw.save_sidecar(sidecar, out_dir, prefix)

    def save_settings(self, out_dir, prefix: str = "") -> None:
        self.settings.save(out_dir, prefix)

    def save_channels(self, out_dir, prefix: str) -> None:
        from . import file_writer as fw

        fw.save_channels(self.channels, out_dir, prefix)

    def save_features(self, feature_arr, out_dir="", prefix: str = "") -> None:
        from . import file_writer as fw

        fw.save_features(feature_arr, out_dir, prefix)

    def reset(self) -> None:
        """Forget everything carried across hops (burst history, Kalman filters, raw and feature normaliser
        histories): the state of a freshly constructed processor."""
        self.engine.reset_state()
        if self._by_len is not None:   # (the twins of other window lengths take their state from whoever ran last)
            for p in self._by_len.values():
                if p is not self:
                    p.engine.reset_state()
            self._active[0] = self
        if self.device_normalizer is not None:
            self.device_normalizer.reset()
        if self._user is not None:
            self._user.reset()
        self.cnt_samples = 0

    # ------------------------------------------------------------------------------------
    def _postprocess_row(self, row: np.ndarray, nan_rows: np.ndarray) -> np.ndarray:
        if self.device_normalizer is not None and not self._norm_in_engine:
            row = self.device_normalizer.process(row)
        if self.feature_normalizer is not None:
            if self.non_psd_indices is not None:
                row = row.copy()
                row[self.non_psd_indices] = self.feature_normalizer.process(row[self.non_psd_indices])
            else:
                row = self.feature_normalizer.process(row)
        if nan_rows.any():
            if len(nan_rows) != len(self.ch_names_used):
                # the reference indexes ch_names_used with the mask over ALL rows (:300)
                raise IndexError("boolean index did not match: NaN handling needs every channel used")
            row = row.copy()
            for ci in np.where(nan_rows)[0]:
                row[self._nan_cols[ci]] = np.nan
        return row

    # -- user-registered features (features/feature_processor.py:52-53,80-82) ----------------------
    @property
    def user_features(self) -> dict:
        return self._user.features if self._user is not None else {}

    @property
    def user_keys(self):
        return self._user.user_keys if self._user is not None else None

    def _user_rows(self, windows) -> np.ndarray:
        first = self._user.user_keys is None
        out = self._user.rows(windows)
        if first and self._user.user_keys is not None:
            self._nan_cols = _LazyNanCols(self.keys, self.ch_names_used)
        return out

    def _host_windows(self, data: np.ndarray, starts) -> "list[np.ndarray]":
        W = self.engine.W_in
        return [np.nan_to_num(np.asarray(data[:, int(s):int(s) + W], dtype=np.float64)) for s in starts]

    def _with_user_columns(self, rows: np.ndarray, user: np.ndarray) -> np.ndarray:
        return self._user.merge(rows, user)

    def process_batch_tapped(self, data: np.ndarray, starts: np.ndarray):
        """One batch through the engine AND the windows its features read: (float32 rows -- normalised when the
        normaliser is attached --, NaN mask, float64 [n, C, W])."""
        eng = self.engine
        if eng.preprocessing_is_identity:
            o, m = eng.process_batch(data, starts, want_nan_mask=True)
            return o, m, np.stack(self._host_windows(data, starts))
        o, m, pre = eng.process_batch(data, starts, want_nan_mask=True, tap=True)
        return o, m, pre.astype(np.float64)

    def _process_batch_user(self, data: np.ndarray, starts: np.ndarray):
        """Engine rows, NaN mask and the user-feature rows of the same hops; the hops go through the engine in chunks
        of ``_user_chunk`` so that the tapped windows stay small."""
        starts = np.asarray(starts, dtype=np.int64)
        if len(starts) == 0:   # (an empty batch is an empty table, as without plugins)
            return (np.empty((0, self.engine.n_outputs), np.float32), np.zeros((0, self.engine.C_in), bool),
                    np.empty((0, len(self._user.user_keys or [])), np.float64))
        outs, masks, users = [], [], []
        for i in range(0, len(starts), self._user_chunk):
            o, m, wins = self.process_batch_tapped(data, starts[i:i + self._user_chunk])
            outs.append(o)
            masks.append(m)
            users.append(self._user_rows(wins))
        return np.concatenate(outs), np.concatenate(masks), np.concatenate(users)

    def _row_dict(self, row: np.ndarray) -> dict:
        """{key: float} in the reference's key order (stream/data_processor.py:238-311 returns a dict).  Built from a template
        that already holds the keys: copying a 10 000-entry dict and overwriting its values costs 0.43 ms where
        ``dict(zip(keys, values))`` -- which grows its table eleven times on the way -- costs 0.72 (the engine's part of a
        256-channel call is 0.36 ms)."""
        keys = self.keys
        tmpl = getattr(self, "_row_template", None)
        if tmpl is None or tmpl[0] is not keys or len(tmpl[1]) != len(keys):
            tmpl = self._row_template = (keys, dict.fromkeys(keys, 0.0), tuple(keys))
        if len(tmpl[1]) != len(tmpl[2]):   # (duplicate keys: nothing a template can hold)
            return dict(zip(keys, row.tolist()))
        d = tmpl[1].copy()
        d.update(zip(tmpl[2], row.tolist()))
        return d

    def _for_length(self, n_samples: int) -> "DataProcessor":
        """The processor for windows of ``n_samples`` (a sampling rate that is not a whole number of samples per segment:
        the reference's generator cuts two lengths, stream/generator.py:41-53, and its DataProcessor takes whatever
        arrives).  One plan per length; what carries over from hop to hop (burst history, Kalman filters, raw-normaliser
        history) is handed from plan to plan when the length changes, and ONE feature normaliser and one set of user
        features serve them all -- the hop-by-hop form of what `Stream.run` does for its ragged runs."""
        if self._by_len is None:
            self._by_len, self._active = {self.engine.W_in: self}, [self]
        p = self._by_len.get(n_samples)
        if p is None:
            if not self._twin_ok:
                raise ValueError(f"expected windows of {self.engine.W_in} samples, got {n_samples}")
            p = DataProcessor(settings=self.settings, channels=self.channels, window=n_samples, **self._ctor)
            if self._norm_in_engine:
                p.device_normalizer = self.device_normalizer
                p.engine.attach_normalizer(self.device_normalizer)
            p.feature_normalizer = self.feature_normalizer
            p._user = self._user
            p._by_len, p._active = self._by_len, self._active
            self._by_len[n_samples] = p
        cur = self._active[0]
        if cur is not p:
            state = cur.engine.export_state()
            if state:
                p.engine.import_state(state)
            p.cnt_samples = cur.cnt_samples
            self._active[0] = p
        return p

    def process(self, data: np.ndarray) -> dict:
        n_samples = np.shape(data)[-1]
        if n_samples != self.engine.W_in or (self._active is not None and self._active[0] is not self):
            p = self._for_length(n_samples)
            if p is not self:
                return p.process(data)
        start_time = time()
        if self._user is not None:
            data = np.asarray(data)
            out, mask, user = self._process_batch_user(data, np.zeros(1, np.int64))
            rows = self._finish_rows(out, mask, self._norm_in_engine)
            row = self._with_user_columns(rows, user)[0]
            if mask[0].any():
                row = self._apply_nan_policy(row[None], mask)[0]
            if self.verbose:
                from . import logger

                logger.info("Last batch took: %.3f seconds to process", time() - start_time)
            return self._row_dict(row)
        out, mask = self.engine.process_window(data, want_nan_mask=True)
        row = self._postprocess_row(out.astype(np.float64), mask)
        if self.verbose:
            from . import logger

            logger.info("Last batch took: %.3f seconds to process", time() - start_time)
        return self._row_dict(row)

    def process_batch(self, data: np.ndarray, starts: np.ndarray, spare_cols: int = 0) -> np.ndarray:
        """data[C_all, T], window start samples -> float64[n, n_features] (same post-processing,
        applied hop by hop because the normaliser is sequential).  ``spare_cols``: the table MAY come back with that
        many extra columns behind the features (HotPathEngine.process_batch_f64) -- the caller checks the shape."""
        if self._user is not None:
            out, mask, user = self._process_batch_user(data, starts)
            rows = self._with_user_columns(self._finish_rows(out, mask, self._norm_in_engine), user)
            return self._apply_nan_policy(rows, mask) if mask.any() else rows
        if self.feature_normalizer is None and (self.device_normalizer is None or self._norm_in_engine):
            # nothing left to do on the host but the NaN policy: conversions pipelined against the device
            rows, mask = self.engine.process_batch_f64(data, starts, want_nan_mask=True, spare_cols=spare_cols)
            return self._apply_nan_policy(rows, mask) if mask.any() else rows
        out, mask = self.engine.process_batch(data, starts, want_nan_mask=True, staged_output=True)
        return self.postprocess_batch(out, mask, normalised=self._norm_in_engine)

    # -- ragged window lengths (a non-integer number of samples per segment): `Stream.run` cuts the hops into
    # consecutive runs of one length, one processor per length; what carries over from hop to hop travels between them
    def ragged_prepare(self) -> None:
        """The feature normaliser is sequential over ALL hops: it runs on the merged table (`ragged_finish`)."""
        if self._norm_in_engine:
            self.engine.attach_normalizer(None)
            self._norm_in_engine = False

    def ragged_state(self):
        """Burst history, Kalman filters ... of the engine (its layout depends on sfreq and the settings, not on the
        window length: nmx_state_export / _import); None when the plan carries nothing."""
        return self.engine.export_state() or None

    def ragged_set_state(self, state) -> None:
        self.engine.import_state(state)

    def ragged_run(self, data: np.ndarray, starts: np.ndarray):
        """One run of equal-length hops -> (float32 engine rows, NaN mask, [pre-processed windows] or None)."""
        eng = self.engine
        if self._user is not None and not eng.preprocessing_is_identity:
            o, m, pre = eng.process_batch(data, starts, want_nan_mask=True, tap=True)
            return o, m, [pre[j].astype(np.float64) for j in range(len(starts))]
        o, m = eng.process_batch(data, starts, want_nan_mask=True)
        return o, m, (self._host_windows(data, starts) if self._user is not None else None)

    def ragged_finish(self, runs) -> np.ndarray:
        """The runs of `ragged_run` in hop order -> the float64 table (normaliser, user columns, NaN policy)."""
        raw = np.concatenate([r[0] for r in runs])
        masks = np.concatenate([r[1] for r in runs])
        if self._user is None:
            return self.postprocess_batch(raw, masks)
        wins = [w for r in runs for w in r[2]]
        user = self._user_rows(wins)   # one set of instances sees every hop in order, like the reference
        rows = self._with_user_columns(self._finish_rows(raw, masks, False), user)
        return self._apply_nan_policy(rows, masks) if masks.any() else rows

    def _finish_rows(self, out: np.ndarray, mask: np.ndarray, normalised: bool) -> np.ndarray:
        """Built-in columns: normalisation (unless it ran inside the engine) and the cast to float64; the NaN policy
        is applied by the caller once the user columns are in place."""
        return self.postprocess_batch(out, np.zeros_like(mask), normalised=normalised)

    def postprocess_batch(self, out: np.ndarray, mask: np.ndarray, normalised: bool = False) -> np.ndarray:
        """Normalisation + NaN policy for engine rows ``out[n, F]`` (hop order); ``normalised``: the
        attached device normaliser already ran inside the engine."""
        from .engine import parallel_cast, table_empty

        if self.device_normalizer is not None and not normalised:
            out = self.device_normalizer.process_batch(out)
        o64 = table_empty(out.shape)
        parallel_cast(o64, out, None, self.engine.lib)
        out = o64
        if self.feature_normalizer is None and not mask.any():
            return out
        if self.feature_normalizer is None:
            return self._apply_nan_policy(out, mask)
        dn, self.device_normalizer = self.device_normalizer, None   # already applied above
        try:
            return np.stack([self._postprocess_row(out[i], mask[i]) for i in range(len(out))])
        finally:
            self.device_normalizer = dn

    def _apply_nan_policy(self, rows: np.ndarray, mask: np.ndarray) -> np.ndarray:
        """Every key that contains the name of a channel whose window held a NaN := NaN (:297-306)."""
        if mask.shape[1] != len(self.ch_names_used):
            raise IndexError("boolean index did not match: NaN handling needs every channel used")
        for ci in np.where(mask.any(axis=0))[0]:
            rows[np.ix_(mask[:, ci], self._nan_cols[ci])] = np.nan
        return rows

"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE.

Build-container only: imports /root/reference through tests/golden/ref_shim.py (which
cannot travel to the GPU box), runs the reference's own classes on seeded inputs and
stores inputs, FIR taps and outputs.  Only data is stored -- no reference source.

    python tests/golden/make_golden.py

Every .npz holds: ``settings_json`` (the reference's ``NMSettings.model_dump()``),
``sfreq``, ``data`` (float64), ``keys``/``values`` per feature class, and the taps the
reference used (designed by oracle.mne_restated injected as mne.filter.create_filter; tap
DESIGN is therefore parity-unpinned against MNE, everything downstream is reference code).
"""

from __future__ import annotations

import json
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shim  # noqa: E402

nm = ref_shim.load_reference()
import py_neuromodulation.features, py_neuromodulation.processing, py_neuromodulation.stream.generator  # noqa: E402
warnings.filterwarnings("ignore")


def synth(C, T, sfreq, seed, dc=True):
    """SURVEY 8(d) synthetic generator: 50 N(0,1) + 10 sin(20 Hz) + 5 sin(70 Hz) + dc_c."""
    rng = np.random.default_rng(seed)
    t = np.arange(T) / sfreq
    x = rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + 5 * np.sin(2 * np.pi * 70 * t)
    if dc:
        x = x + rng.uniform(-500, 500, size=(C, 1))
    return x


def dump(settings):
    return json.dumps(settings.model_dump())


def pack(d: dict, prefix: str, out: dict):
    out[prefix + "_keys"] = np.array(list(d.keys()))
    out[prefix + "_values"] = np.array([float(v) for v in d.values()], dtype=np.float64)


def all_estimators(s):
    for name in ("fft_settings", "welch_settings", "stft_settings"):
        for e in ("mean", "median", "std", "max"):
            setattr(s[name].features, e, True)


def sharpwave_all(s):
    sw = s.sharpwave_analysis_settings
    sw.sharpwave_features.enable_all()
    feats = list(type(sw.sharpwave_features).model_fields.keys())
    sw.estimator.mean = list(feats)
    sw.estimator.median = ["prominence", "interval"]
    sw.estimator.max = ["prominence", "sharpness", "rise_steepness"]
    sw.estimator.min = ["decay_time", "sharpness"]
    sw.estimator.var = ["interval", "width"]


def feature_case(name, sfreq, data, mutate):
    s = nm.NMSettings.get_default()
    s.features.enable_all()
    for f in ("fooof", "nolds", "coherence", "mne_connectivity", "bispectrum"):
        setattr(s.features, f, False)
    mutate(s)
    s = s.validate()
    ch_names = [f"ch{i}" for i in range(data.shape[0])]
    out = {"settings_json": dump(s), "sfreq": sfreq, "data": data, "ch_names": np.array(ch_names)}
    pack(nm.features.Hjorth(s, ch_names, sfreq).calc_feature(data), "hjorth", out)
    pack(nm.features.Raw(s, ch_names, sfreq).calc_feature(data), "raw", out)
    pack(nm.features.LineLength(s, ch_names, sfreq).calc_feature(data), "linelength", out)
    pack(nm.features.FFT(s, ch_names, sfreq).calc_feature(data), "fft", out)
    pack(nm.features.Welch(s, ch_names, sfreq).calc_feature(data), "welch", out)
    pack(nm.features.STFT(s, ch_names, sfreq).calc_feature(data), "stft", out)
    bp = nm.features.BandPower(s, ch_names, sfreq)
    out["bank_taps"] = bp.bandpass_filter.filter_bank
    out["bank_filtered"] = bp.bandpass_filter.filter_data(data)[:2]
    pack(bp.calc_feature(data), "bandpass", out)
    sw = nm.features.SharpwaveAnalyzer(s, ch_names, sfreq)
    for i, (_, taps) in enumerate(sw.list_filter):
        out[f"sw_taps_{i}"] = taps
    pack(sw.calc_feature(data), "sharpwave", out)
    out["sw_filtered"] = sw.filtered_data[:2]
    bu = nm.features.Bursts(s, ch_names, sfreq)
    out["bursts_taps"] = bu.bandpass_filter.filter_bank
    pack(bu.calc_feature(data), "bursts", out)
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "values" in k})


def case_feat_1k():
    def mut(s):
        all_estimators(s)
        sharpwave_all(s)
        for f in ("activity", "mobility", "complexity"):
            setattr(s.bandpass_filter_settings.bandpower_features, f, True)
    feature_case("feat_1k", 1000, synth(4, 1000, 1000, 11), mut)

    def mut2(s):
        mut(s)
        for n in ("fft_settings", "welch_settings", "stft_settings"):
            s[n].log_transform = False
        s.bandpass_filter_settings.log_transform = False
        s.sharpwave_analysis_settings.apply_estimator_between_peaks_and_troughs = False
    feature_case("feat_1k_nolog", 1000, synth(3, 1000, 1000, 12, dc=False), mut2)


def case_feat_2k():
    def mut(s):
        s.segment_length_features_ms = 1000
        s.frequency_ranges_hz = {
            "theta": [4, 8], "alpha": [8, 12], "low_beta": [13, 20], "high_beta": [20, 35],
            "low_gamma": [60, 80], "high_gamma": [90, 200], "HFA": [200, 400],
            "broadband": [4, 400]}
        s.bandpass_filter_settings.segment_lengths_ms["broadband"] = 1000
        s.stft_settings.windowlength_ms = 500
        s.bursts_settings.frequency_bands = ["low_beta", "high_beta"]
    feature_case("feat_2k", 2000, synth(2, 2000, 2000, 13), mut)


def case_special_rows():
    rng = np.random.default_rng(14)
    data = np.vstack([np.zeros(1000), np.full(1000, 3.25), rng.random(1000),
                      np.sin(2 * np.pi * 20 * np.arange(1000) / 1000) + rng.random(1000)])

    def mut(s):
        all_estimators(s)
    feature_case("feat_special_rows", 1000, data, mut)


def case_bursts_sequence():
    sfreq, C, T = 1000, 2, 6000
    s = nm.NMSettings.get_default()
    s.bursts_settings.time_duration_s = 2
    s.bursts_settings.frequency_bands = ["low_beta", "high_beta"]
    s = s.validate()
    rng = np.random.default_rng(15)
    t = np.arange(T) / sfreq
    amp = 1 + 0.8 * np.sin(2 * np.pi * 0.7 * t)
    data = rng.standard_normal((C, T)) * 20 + 30 * amp * np.sin(2 * np.pi * 18 * t)
    ch_names = [f"ch{i}" for i in range(C)]
    bu = nm.features.Bursts(s, ch_names, sfreq)
    gen = nm.stream.generator.RawDataGenerator(data, sfreq, s.sampling_rate_features_hz,
                                               s.segment_length_features_ms)
    rows, keys = [], None
    for _, w in gen:
        d = bu.calc_feature(w)
        keys = list(d.keys())
        rows.append([float(v) for v in d.values()])
    out = {"settings_json": dump(s), "sfreq": sfreq, "data": data, "ch_names": np.array(ch_names),
           "bursts_taps": bu.bandpass_filter.filter_bank, "keys": np.array(keys),
           "values": np.array(rows)}
    np.savez_compressed(HERE / "bursts_sequence.npz", **out)
    print("bursts_sequence", out["values"].shape)


def case_c5_degenerate():
    """BASELINE config[4] (30 kHz, 17 ms windows, 1 kHz feature rate) where the reference degenerates:
    * Bursts: samples_overlap = int(sfreq * seg_s / feat_hz) = int(0.51) = 0, and `filtered_data[:, :, -0:]` is the
      WHOLE window -- every hop appends all 510 envelope samples to the percentile buffer (features/bursts.py:81-85,
      155-166); recorded over 40 hops across the ring overflow (time_duration_s = 0.5 -> 15 000 samples);
    * Welch: nperseg = sfreq = 30 000 > the window; scipy shrinks the segment to the window while the reference's
      band indices keep the 1 Hz grid (features/oscillatory.py:136-144): bands below bin 256 read OTHER frequencies,
      any band beyond raises IndexError.  Both recorded."""
    sfreq, C = 30000, 2
    base = nm.NMSettings.get_default()
    base.frequency_ranges_hz = {"gamma": [60, 200], "HFA": [200, 500], "MUA": [500, 3000]}
    base.sampling_rate_features_hz = 1000
    base.segment_length_features_ms = 17
    for name in ("fft_settings", "welch_settings", "stft_settings"):
        base[name].windowlength_ms = 17
    base.bandpass_filter_settings.segment_lengths_ms = {"gamma": 17, "HFA": 10, "MUA": 5}
    base.bursts_settings.frequency_bands = ["HFA", "MUA"]
    base.bursts_settings.time_duration_s = 0.5
    s = base.validate()
    W, hop, nh = 510, 30, 40
    T = W + (nh - 1) * hop
    rng = np.random.default_rng(55)
    t = np.arange(T) / sfreq
    amp = 1 + 0.8 * np.sin(2 * np.pi * 40 * t)
    data = rng.standard_normal((C, T)) * 20 + 30 * amp * np.sin(2 * np.pi * 900 * t) + 15 * np.sin(2 * np.pi * 300 * t)
    ch_names = [f"ch{i}" for i in range(C)]
    out = {"settings_json": dump(s), "sfreq": sfreq, "data": data, "ch_names": np.array(ch_names), "W": W, "hop": hop}
    bu = nm.features.Bursts(s, ch_names, sfreq)
    assert bu.samples_overlap == 0
    rows, keys = [], None
    for i in range(nh):
        d = bu.calc_feature(data[:, i * hop:i * hop + W])
        keys = list(d.keys())
        rows.append([float(v) for v in d.values()])
    out["bursts_taps"] = bu.bandpass_filter.filter_bank
    out["bursts_keys"] = np.array(keys)
    out["bursts_values"] = np.array(rows)
    # Welch with the three bands: IndexError; with the one band whose indices stay inside the shrunk spectrum: values
    try:
        nm.features.Welch(s, ch_names, sfreq).calc_feature(data[:, :W])
        out["welch_error"] = ""
    except Exception as e:   # noqa: BLE001
        out["welch_error"] = type(e).__name__
    s2 = nm.NMSettings.get_default()
    s2.frequency_ranges_hz = {"gamma": [60, 200]}
    s2.sampling_rate_features_hz = 1000
    s2.segment_length_features_ms = 17
    for name in ("fft_settings", "welch_settings", "stft_settings"):
        s2[name].windowlength_ms = 17
    s2.bandpass_filter_settings.segment_lengths_ms = {"gamma": 17}
    s2.bursts_settings.frequency_bands = ["gamma"]
    s2 = s2.validate()
    out["welch1_settings_json"] = dump(s2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pack(nm.features.Welch(s2, ch_names, sfreq).calc_feature(data[:, :W]), "welch1", out)
    np.savez_compressed(HERE / "c5_degenerate.npz", **out)
    print("c5_degenerate", out["bursts_values"].shape, out["welch_error"], len(out["welch1_keys"]))


def case_sharpwave_tests():
    """Inputs of the reference's tests/test_sharpwave.py (impulses, sines)."""
    sfreq = 1000
    s = nm.NMSettings.get_default()
    sharpwave_all(s)
    s = s.validate()
    W = 1000
    rows = []
    for height in (1, 2, 3, 4):
        x = np.zeros(W)
        x[100::200] = height
        rows.append(x)
    for spacing in (100, 200, 300, 400):
        x = np.zeros(W)
        x[50::spacing] = 1.0
        rows.append(x)
    t = np.arange(W) / sfreq
    rng = np.random.default_rng(16)
    rows.append(np.sin(2 * np.pi * 5 * t) + 0.1 * rng.standard_normal(W))
    rows.append(np.sin(2 * np.pi * 15 * t) + 0.1 * rng.standard_normal(W))
    rows.append(np.round(5 * np.sin(2 * np.pi * 9 * t)))  # plateaus
    data = np.vstack(rows)
    ch_names = [f"ch{i}" for i in range(data.shape[0])]
    sw = nm.features.SharpwaveAnalyzer(s, ch_names, sfreq)
    out = {"settings_json": dump(s), "sfreq": sfreq, "data": data, "ch_names": np.array(ch_names)}
    for i, (_, taps) in enumerate(sw.list_filter):
        out[f"sw_taps_{i}"] = taps
    pack(sw.calc_feature(data), "sharpwave", out)
    np.savez_compressed(HERE / "sharpwave_tests.npz", **out)
    print("sharpwave_tests", out["sharpwave_values"].shape)


def _run_stream(data, sfreq, s, channels=None, line_noise=50):
    st = nm.Stream(sfreq=sfreq, data=data if channels is None else None, channels=channels,
                   settings=s, line_noise=line_noise, verbose=False)
    with tempfile.TemporaryDirectory() as td:
        df = st.run(data=data, out_dir=td, save_csv=False)
    return st, df


def case_pipeline():
    """README demo shape (README.rst:73-86): 5 ch x 10000 random, 3 Hz features."""
    np.random.seed(0)
    data = np.random.random([5, 10000])
    out = {"sfreq": 1000, "data": data}
    for tag, mutate in {
        "default": lambda s: None,                      # notch+reref+zscore (notch taps unpinned)
        "reref_nonorm": lambda s: (setattr(s, "preprocessing", ["re_referencing"]),
                                   setattr(s.postprocessing, "feature_normalization", False)),
        "nopre_norm": lambda s: setattr(s, "preprocessing", []),
        # the un-normalised features of "default": lets the tests check features and normaliser separately
        "default_nonorm": lambda s: setattr(s.postprocessing, "feature_normalization", False),
    }.items():
        s = nm.NMSettings.get_default()
        s.features.bandpass_filter = True
        s.features.stft = True
        s.sampling_rate_features_hz = 3
        mutate(s)
        st, df = _run_stream(data, 1000, s)
        out[f"{tag}_settings_json"] = dump(st.settings)
        out[f"{tag}_columns"] = np.array(list(df.columns))
        out[f"{tag}_values"] = df.to_numpy(dtype=np.float64)
        out[f"{tag}_channels_json"] = json.dumps(st.channels.to_dict("list"))
        print("pipeline", tag, df.shape)
    np.savez_compressed(HERE / "pipeline_readme.npz", **out)


def _small_settings():
    s = nm.NMSettings.get_default()
    s.reset()
    s.features.fft = True
    s.features.raw_hjorth = True
    s.features.linelength = True
    s.preprocessing = ["re_referencing"]
    s.postprocessing.feature_normalization = False
    return s


def case_nan_and_channels():
    """tests/test_nan_values.py (NaN policy) and bad/target/bipolar channel handling."""
    import pandas as pd

    rng = np.random.default_rng(17)
    # (1) NaN policy with the default channel table (every channel used; the reference's NaN
    # mask indexing requires that, data_processor.py:253,300)
    data = rng.standard_normal((3, 3000)) * 10
    data[1, 1500:1510] = np.nan
    st, df = _run_stream(data, 1000, _small_settings())
    out = {"settings_json": dump(st.settings), "sfreq": 1000, "nan_data": data,
           "nan_columns": np.array(list(df.columns)), "nan_values": df.to_numpy(dtype=np.float64),
           "nan_channels_json": json.dumps(st.channels.to_dict("list"))}
    # (2) mixed channel table: bipolar refs, a bad channel, an unused target channel
    data = rng.standard_normal((6, 3000)) * 10
    ch = pd.DataFrame({
        "name": ["LFP_0", "LFP_1", "ECOG_0", "ECOG_1", "ECOG_2", "MOV"],
        "rereference": ["LFP_1", "LFP_0", "average", "average", "average", "None"],
        "used": [1, 1, 1, 1, 1, 0],
        "target": [0, 0, 0, 0, 0, 1],
        "type": ["seeg", "seeg", "ecog", "ecog", "ecog", "misc"],
        "status": ["good", "good", "good", "bad", "good", "good"],
        "new_name": ["LFP_0-LFP_1", "LFP_1-LFP_0", "ECOG_0-avgref", "ECOG_1-avgref",
                     "ECOG_2-avgref", "MOV"],
    })
    st, df = _run_stream(data, 1000, _small_settings(), channels=ch)
    rr = nm.processing.ReReferencer(1000, ch)
    out.update({"mix_data": data, "mix_columns": np.array(list(df.columns)),
                "mix_values": df.to_numpy(dtype=np.float64),
                "mix_channels_json": json.dumps(ch.to_dict("list")),
                "mix_ref_matrix": rr.ref_matrix})
    np.savez_compressed(HERE / "pipeline_nan_channels.npz", **out)
    print("pipeline_nan_channels", out["nan_values"].shape, out["mix_values"].shape)


def case_schedule():
    out = {}
    for tag, (T, sfreq, fh, seg) in {
        "a": (10000, 1000, 3, 1000), "b": (10000, 1000, 10, 1000), "c": (3000, 1000, 200, 1000),
        "d": (12000, 1111.111, 10, 1000), "e": (9000, 2000, 7.5, 500),
    }.items():
        data = np.zeros((1, T))
        gen = nm.stream.generator.RawDataGenerator(data, sfreq, fh, seg)
        starts, lens, times = [], [], []
        base = data.ctypes.data
        for ts, w in gen:
            starts.append((w.ctypes.data - base) // 8)
            lens.append(w.shape[1])
            times.append(float(np.ceil(ts[-1] * 1000 + 1)))
        out[f"{tag}_params"] = np.array([T, sfreq, fh, seg], dtype=np.float64)
        out[f"{tag}_starts"] = np.array(starts)
        out[f"{tag}_lens"] = np.array(lens)
        out[f"{tag}_times"] = np.array(times)
    np.savez_compressed(HERE / "schedule.npz", **out)
    print("schedule", {k: len(v) for k, v in out.items() if k.endswith("starts")})


def case_notch():
    """NotchFilter.process = reference glue over the RESTATED _overlap_add_filter (unpinned)."""
    out = {}
    for sfreq in (1000, 2000):
        nf = nm.processing.NotchFilter(sfreq, line_noise=50)
        x = synth(2, sfreq, sfreq, 18)
        out[f"taps_{sfreq}"] = nf.filter_bank
        out[f"x_{sfreq}"] = x
        out[f"y_{sfreq}"] = nf.process(x)
    np.savez_compressed(HERE / "notch_unpinned.npz", **out)
    print("notch", [k for k in out])


def case_bandpower_kalman():
    """BandPower with kalman_filter over consecutive hops (bandpower.py:147-163,188-189): the
    reference's own filterpy-derived KalmanFilter runs per (channel, band); "high_beta" is left out
    of kalman_filter_settings.frequency_bands so filtered and unfiltered bands are mixed."""
    sfreq, C, T = 1000, 2, 5000
    s = nm.NMSettings.get_default().reset()
    s.features.bandpass_filter = True
    s.bandpass_filter_settings.kalman_filter = True
    s.bandpass_filter_settings.bandpower_features.mobility = True
    s.kalman_filter_settings.frequency_bands = ["theta", "alpha", "low_beta"]
    s = s.validate()
    rng = np.random.default_rng(21)
    t = np.arange(T) / sfreq
    amp = 1 + 0.9 * np.sin(2 * np.pi * 0.5 * t)
    data = rng.standard_normal((C, T)) * 10 + 25 * amp * np.sin(2 * np.pi * 10 * t)
    ch_names = [f"ch{i}" for i in range(C)]
    bp = nm.features.BandPower(s, ch_names, sfreq)
    gen = nm.stream.generator.RawDataGenerator(data, sfreq, s.sampling_rate_features_hz,
                                               s.segment_length_features_ms)
    rows, keys = [], None
    for _, w in gen:
        d = bp.calc_feature(w)
        keys = list(d.keys())
        rows.append([float(v) for v in d.values()])
    out = {"settings_json": dump(s), "sfreq": sfreq, "data": data, "ch_names": np.array(ch_names),
           "bandpass_taps": bp.bandpass_filter.filter_bank, "keys": np.array(keys),
           "values": np.array(rows)}
    np.savez_compressed(HERE / "bandpower_kalman.npz", **out)
    print("bandpower_kalman", out["values"].shape)


def case_preprocessing_filter():
    """PreprocessingFilter.process (processing/filter_preprocessing.py:44-94) with the default four
    filters and with a subset: reference glue (MNEFilter tiling + fftconvolve 'same', chained) over
    the restated tap design; the taps are stored next to the output."""
    sfreq = 1000
    out = {"sfreq": sfreq}
    for tag, mutate in (("all", lambda s: None),
                        ("two", lambda s: (setattr(s.preprocessing_filter, "bandstop_filter", False),
                                           setattr(s.preprocessing_filter, "highpass_filter", False)))):
        s = nm.NMSettings.get_default()
        mutate(s)
        s.preprocessing = ["preprocessing_filter"]
        s = s.validate()
        pf = nm.processing.PreprocessingFilter(s, sfreq)
        x = synth(3, 1000, sfreq, 31 + len(tag))
        y = pf.process(x)
        out[f"{tag}_settings_json"] = dump(s)
        out[f"{tag}_x"] = x
        out[f"{tag}_y"] = y
        out[f"{tag}_n_filters"] = len(pf.filters)
        for i, f in enumerate(pf.filters):
            out[f"{tag}_taps_{i}"] = f.filter_bank[0]
    np.savez_compressed(HERE / "preprocessing_filter.npz", **out)
    print("preprocessing_filter", {k: np.shape(v) for k, v in out.items() if k.endswith("_y")})


def case_raw_normalizer():
    """RawNormalizer.process over consecutive windows (processing/normalization.py:31-116, type "raw"):
    zscore and mean, a short history (0.5 s) so that the N - 1 trim is exercised, clip on/off."""
    sfreq, C, T = 1000, 2, 2200
    out = {"sfreq": sfreq, "stride": 4}   # every 4th sample of each output window is stored
    data = synth(C, T, sfreq, 41)
    out["data"] = data
    for tag, method, clip in (("zscore", "zscore", 3), ("mean", "mean", 3), ("zscore_noclip", "zscore", 0)):
        s = nm.NMSettings.get_default()
        s.raw_normalization_settings.normalization_time_s = 0.5
        s.raw_normalization_settings.normalization_method = method
        s.raw_normalization_settings.clip = clip
        s.preprocessing = ["raw_normalization"]
        s = s.validate()
        rn = nm.processing.RawNormalizer(sfreq, s)
        gen = nm.stream.generator.RawDataGenerator(data, sfreq, s.sampling_rate_features_hz,
                                                   s.segment_length_features_ms)
        rows = [np.array(rn.process(np.array(w, dtype=np.float64))) for _, w in gen]
        out[f"{tag}_settings_json"] = dump(s)
        out[f"{tag}_y"] = np.stack(rows)[:, :, ::4]
    np.savez_compressed(HERE / "raw_normalizer.npz", **out)
    print("raw_normalizer", {k: np.shape(v) for k, v in out.items() if k.endswith("_y")})


def case_norm_methods():
    """Every normalization_method the reference lists (processing/normalization.py:57-70) through its own
    Normalizer, scikit-learn included: FeatureNormalizer over 160 hops x 9 columns (N = 50: trims, NaN cells, a
    constant column, ties) and RawNormalizer over 14 windows of 3 channels (N = 700 samples)."""
    rng = np.random.default_rng(99)
    n, F = 160, 9
    rows = rng.standard_normal((n, F)) * rng.uniform(0.05, 20, F) + rng.uniform(-30, 30, F)
    rows[:, 2] = 1.25                                      # constant column
    rows[:, 3] = np.round(rows[:, 3])                      # ties (repeated quantiles)
    rows[rng.integers(0, n, 12), rng.integers(4, 7, 12)] = np.nan
    rows[:, 7] = np.exp(rows[:, 7] / 20)                   # skewed, positive
    out = {"rows": rows}
    for method in nm.NMSettings.list_normalization_methods():
        for clip in (3, 0):
            s = nm.NMSettings.get_default()
            s.sampling_rate_features_hz = 10
            s.feature_normalization_settings.normalization_time_s = 5
            s.feature_normalization_settings.normalization_method = method
            s.feature_normalization_settings.clip = clip
            fnorm = nm.processing.FeatureNormalizer(s)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out[f"feature_{method}_clip{clip}"] = np.stack([np.array(fnorm.process(r.copy())) for r in rows])
    sfreq, C, T = 1000, 3, 2300
    data = synth(C, T, sfreq, 43)
    data[1] = np.round(data[1] / 5) * 5                    # coarse quantisation: many equal samples
    out["raw_data"] = data
    for method in nm.NMSettings.list_normalization_methods():
        s = nm.NMSettings.get_default()
        s.raw_normalization_settings.normalization_time_s = 0.7
        s.raw_normalization_settings.normalization_method = method
        s.raw_normalization_settings.clip = 3
        s.preprocessing = ["raw_normalization"]
        s = s.validate()
        rn = nm.processing.RawNormalizer(sfreq, s)
        gen = nm.stream.generator.RawDataGenerator(data, sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[f"raw_{method}"] = np.stack([np.array(rn.process(np.array(w, dtype=np.float64))) for _, w in gen])[:, :, ::4]
    np.savez_compressed(HERE / "norm_methods.npz", **out)
    print("norm_methods", sorted(k for k in out if k.startswith(("feature_", "raw_"))))


def case_short_windows():
    """Windows SHORTER than the nominal spectral segment: scipy.signal.welch / stft shrink nperseg to the window
    (with a warning) while the reference keeps the band indices of the nominal frequency grid
    (features/oscillatory.py:136-144, 201-210): it then returns features of the wrong bins, or raises IndexError
    when a band index lies beyond the shorter spectrum.  Both behaviours are recorded here."""
    out = {}
    cases = {"a": (750, 187, 250, {"theta": [4, 8], "alpha": [8, 12], "high_beta": [20, 35]}),
             "b": (1000, 500, 500, None),
             "c": (512, 128, 250, {"alpha": [8, 12], "low_beta": [13, 20], "high_gamma": [90, 200]}),
             "d": (2000, 600, 1000, None)}
    for tag, (sfreq, W, wl_ms, bands) in cases.items():
        s = nm.NMSettings.get_default()
        if bands is not None:
            s.frequency_ranges_hz = bands
        s.segment_length_features_ms = int(1000 * W / sfreq) + 1
        for name in ("fft_settings", "welch_settings", "stft_settings"):
            s[name].windowlength_ms = min(wl_ms, s.segment_length_features_ms)
        all_estimators(s)
        x = synth(3, W, sfreq, 70 + ord(tag), dc=False)
        out[f"{tag}_settings_json"] = dump(s)
        out[f"{tag}_sfreq"] = sfreq
        out[f"{tag}_data"] = x
        ch = ["c0", "c1", "c2"]
        for fam, cls in (("fft", nm.features.FFT), ("welch", nm.features.Welch), ("stft", nm.features.STFT)):
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    d = cls(s, ch, sfreq).calc_feature(x)
                pack(d, f"{tag}_{fam}", out)
                out[f"{tag}_{fam}_error"] = ""
            except Exception as e:   # noqa: BLE001 -- the error class is the recorded behaviour
                out[f"{tag}_{fam}_error"] = type(e).__name__
            print("short_windows", tag, fam, out[f"{tag}_{fam}_error"] or len(out[f"{tag}_{fam}_keys"]))
    np.savez_compressed(HERE / "short_windows.npz", **out)


def case_resample_quirk():
    """Recordings that are NOT sampled at raw_resampling_settings.resample_freq_hz (1000 Hz) under the default
    pre-processing: the reference resamples every window (processing/resample.py:42-60) but keeps building
    notch and features with the RAW rate (stream/data_processor.py:55,68,80).  Down- (2 kHz) and up-sampling
    (250 Hz) through the reference's own Stream.run; restated mne.filter.resample (parity unpinned vs MNE)."""
    out = {}
    for tag, sf in (("down_2k", 2000), ("up_250", 250)):
        rng = np.random.default_rng(50 + sf)
        T = int(2 * sf + 6 * sf / 10)
        t = np.arange(T) / sf
        data = (rng.standard_normal((3, T)) * 20 + 30 * np.sin(2 * np.pi * 50 * t) + 25 * np.sin(2 * np.pi * 17 * t)
                + rng.uniform(-100, 100, (3, 1)))
        s = nm.NMSettings.get_default()
        s.features.bandpass_filter = True
        s.features.stft = True
        s.postprocessing.feature_normalization = False
        st, df = _run_stream(data, sf, s)
        out[f"{tag}_sfreq"] = sf
        out[f"{tag}_data"] = data
        out[f"{tag}_settings_json"] = dump(st.settings)
        out[f"{tag}_columns"] = np.array(list(df.columns))
        out[f"{tag}_values"] = df.to_numpy(dtype=np.float64)
        out[f"{tag}_channels_json"] = json.dumps(st.channels.to_dict("list"))
        print("resample_quirk", tag, df.shape)
    np.savez_compressed(HERE / "resample_quirk.npz", **out)


def case_output_files():
    """The files the reference's OWN Stream.run leaves behind (stream/stream.py:229,319-343,426-453;
    utils/file_writer.py:53-118 MsgPackFileWriter unmodified): names, the per-interval msgpack layout, CSV
    header and values, sidecar, channels table and the top-level keys of the settings YAML.  Stored as data
    (strings / arrays) -- the GPU box compares what the engine's Stream.run writes with these."""
    import msgpack
    import yaml

    s = nm.NMSettings.get_default()
    s.reset()
    s.features.fft = True
    s.features.raw_hjorth = True
    s.preprocessing = ["re_referencing"]
    rng = np.random.default_rng(3)
    data = rng.standard_normal((3, 4300))
    st = nm.Stream(sfreq=1000.0, data=data, settings=s, sampling_rate_features_hz=10, verbose=False)
    with tempfile.TemporaryDirectory() as td:
        df = st.run(data, out_dir=td, experiment_name="sub7", save_csv=True, save_interval=10,
                    delete_ind_batch_files_after_stream=False)
        out_dir = Path(td) / "sub7"
        names = sorted(p.name for p in out_dir.iterdir())
        packs = sorted(out_dir.glob("sub7-*.msgpack"), key=lambda p: int(p.stem.split("-")[1]))
        rows_per_file, first = [], None
        for p in packs:
            with open(p, "rb") as f:
                d = msgpack.unpack(f)
            rows_per_file.append(len(d))
            if first is None:
                first = d
        csv_text = (out_dir / "sub7_FEATURES.csv").read_text()
        out = {"data": data, "settings_json": dump(st.settings), "file_names": np.array(names),
               "df_columns": np.array(list(df.columns)), "df_values": df.to_numpy(dtype=np.float64),
               "df_dtypes": np.array([str(t) for t in df.dtypes]),
               "msgpack_rows_per_file": np.array(rows_per_file),
               "msgpack_first_keys": np.array(list(first[0].keys())),
               "msgpack_first_types": np.array([type(v).__name__ for v in first[0].values()]),
               "csv_header": csv_text.splitlines()[0], "csv_n_lines": len(csv_text.splitlines()),
               "sidecar_json": (out_dir / "sub7_SIDECAR.json").read_text(),
               "channels_csv": (out_dir / "sub7_channels.csv").read_text(),
               "settings_yaml_keys": np.array(list(yaml.safe_load((out_dir / "sub7_SETTINGS.yaml").read_text()).keys()))}
    # default call: the per-interval files are deleted after the run (delete_ind_batch_files_after_stream=True)
    with tempfile.TemporaryDirectory() as td:
        st2 = nm.Stream(sfreq=1000.0, data=data, settings=s, sampling_rate_features_hz=10, verbose=False)
        st2.run(data, out_dir=td, experiment_name="sub7")
        out["file_names_default_call"] = np.array(sorted(p.name for p in (Path(td) / "sub7").iterdir()))
    np.savez_compressed(HERE / "output_files.npz", **out)
    print("output_files", names, rows_per_file)


def case_ragged_bursts():
    """A sampling rate that is not an integer (tests/test_feature_sampling_rates.py: 1111.111 Hz): the generator cuts
    windows of 1111 AND 1112 samples.  Bursts append the last int(sfreq * seg_s / feat_hz) samples of whatever window
    arrives to ONE history, the Kalman filters of the band powers run on: state that crosses window lengths."""
    rng = np.random.default_rng(5150)
    sfreq = 1111.111
    T = 9000
    t = np.arange(T) / sfreq
    data = rng.standard_normal((2, T)) * 20 + 15 * np.sin(2 * np.pi * 18 * t) * (np.sin(2 * np.pi * 0.7 * t) > 0)
    s = nm.NMSettings.get_default()
    s.reset()
    s.features.fft = True
    s.features.bursts = True
    s.features.bandpass_filter = True
    s.bandpass_filter_settings.kalman_filter = True
    s.kalman_filter_settings.frequency_bands = ["theta", "low_beta"]
    s.bursts_settings.time_duration_s = 2
    s.preprocessing = []
    s.postprocessing.feature_normalization = False
    st, df = _run_stream(data, sfreq, s)
    gen = nm.stream.generator.RawDataGenerator(data, sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    lens = [b.shape[1] for _, b in gen]
    assert len(set(lens)) == 2, lens
    bp = nm.features.BandPower(s, ["ch0", "ch1"], sfreq // 1)
    out = {"sfreq": sfreq, "data": data, "settings_json": dump(st.settings), "columns": np.array(list(df.columns)),
           "values": df.to_numpy(dtype=np.float64), "window_lengths": np.array(lens),
           "channels_json": json.dumps(st.channels.to_dict("list")),
           "bank_taps": np.asarray(bp.bandpass_filter.filter_bank)}
    np.savez_compressed(HERE / "ragged_bursts.npz", **out)
    print("ragged_bursts", df.shape, sorted(set(lens)))


def case_long_windows():
    """Windows beyond two former kernel limits, at 7 kHz with 1 s windows (7000 samples) and 10 Hz features (700-sample
    hops): (sw) sharp waves on windows of more than 4092 samples -- the default settings, and every feature with
    median / var estimators and the estimator applied per polarity; the second channel is coarsely quantised
    (plateaus).  (rn) raw_normalization with the order-statistic methods where window + hop = 7700 > 6484 samples
    (0.9 s history: the N - 1 trim acts every hop), in front of fft + return_raw.  The reference's own Stream.run."""
    rng = np.random.default_rng(707)
    sfreq = 7000
    T = 7000 + 9 * 700
    t = np.arange(T) / sfreq
    data = np.cumsum(rng.standard_normal((2, T)), axis=1) * 0.05 + rng.standard_normal((2, T)) * 0.5
    data += 6 * np.sin(2 * np.pi * 11 * t) + 3 * np.sin(2 * np.pi * 47 * t + 1.0)
    data[1] = np.round(data[1] * 1.5) / 1.5
    out = {"sfreq": sfreq, "data": data.astype(np.float32)}   # (float32-exact inputs: half the fixture)
    data = out["data"].astype(np.float64)

    def base():
        s = nm.NMSettings.get_default()
        s.reset()
        s.preprocessing = []
        s.postprocessing.feature_normalization = False
        return s

    for tag in ("sw_default", "sw_all"):
        s = base()
        s.features.sharpwave_analysis = True
        if tag == "sw_all":
            sharpwave_all(s)
            s.sharpwave_analysis_settings.apply_estimator_between_peaks_and_troughs = False
        st, df = _run_stream(data, sfreq, s)
        sw = nm.features.SharpwaveAnalyzer(st.settings, ["ch0", "ch1"], sfreq)
        out.update({f"{tag}_settings_json": dump(st.settings), f"{tag}_columns": np.array(list(df.columns)),
                    f"{tag}_values": df.to_numpy(dtype=np.float64),
                    f"{tag}_channels_json": json.dumps(st.channels.to_dict("list"))})
        for i, (_, taps) in enumerate(sw.list_filter):
            out[f"sw_taps_{i}"] = np.asarray(taps)
        print("long_windows", tag, df.shape)
    for method in ("median", "zscore-median", "robust", "minmax"):
        s = base()
        s.features.fft = True
        s.features.return_raw = True
        s.preprocessing = ["raw_normalization"]
        s.raw_normalization_settings.normalization_time_s = 0.9
        s.raw_normalization_settings.normalization_method = method
        s.raw_normalization_settings.clip = 3
        st, df = _run_stream(data + 20.0, sfreq, s)    # (offset: "median" divides by the median)
        tag = "rn_" + method.replace("-", "_")
        out.update({f"{tag}_settings_json": dump(st.settings), f"{tag}_columns": np.array(list(df.columns)),
                    f"{tag}_values": df.to_numpy(dtype=np.float64),
                    f"{tag}_channels_json": json.dumps(st.channels.to_dict("list"))})
        print("long_windows", tag, df.shape)
    np.savez_compressed(HERE / "long_windows.npz", **out)


def case_ragged_rawnorm():
    """raw_normalization on a stream with ragged window lengths (1111.111 Hz: windows of 1111 and 1112 samples): ONE
    sample history -- the whole first window, then the last int(sfreq / feat_hz) samples of every later one, trimmed to
    N - 1 = int(time_s * sfreq) - 1 (processing/normalization.py:31-116) -- whatever the window length; zscore with a
    1.5 s history (trimmed every hop) and the order-statistic median on a 3 s one."""
    rng = np.random.default_rng(8112)
    sfreq, T = 1111.111, 7000
    t = np.arange(T) / sfreq
    data = rng.standard_normal((2, T)) * 12 + 30 + 8 * np.sin(2 * np.pi * 9 * t)
    out = {"sfreq": sfreq, "data": data}
    for tag, method, ts in (("zscore", "zscore", 1.5), ("median", "median", 3.0)):
        s = nm.NMSettings.get_default()
        s.reset()
        s.features.fft = True
        s.features.return_raw = True
        s.features.raw_hjorth = True
        s.preprocessing = ["raw_normalization"]
        s.raw_normalization_settings.normalization_time_s = ts
        s.raw_normalization_settings.normalization_method = method
        s.raw_normalization_settings.clip = 3
        s.postprocessing.feature_normalization = False
        st, df = _run_stream(data, sfreq, s)
        out.update({f"{tag}_settings_json": dump(st.settings), f"{tag}_columns": np.array(list(df.columns)),
                    f"{tag}_values": df.to_numpy(dtype=np.float64),
                    f"{tag}_channels_json": json.dumps(st.channels.to_dict("list"))})
        print("ragged_rawnorm", tag, df.shape)
    gen = nm.stream.generator.RawDataGenerator(data, sfreq, 10, 1000)
    out["window_lengths"] = np.array([b.shape[1] for _, b in gen])
    assert len(set(out["window_lengths"].tolist())) == 2
    np.savez_compressed(HERE / "ragged_rawnorm.npz", **out)


def case_user_features():
    """User-registered NMFeature plugins (features/feature_processor.py:52-53,90-108; the plugin of
    examples/plot_2_example_add_feature.py is tests/user_plugins.ChannelMean): the reference's own Stream.run with
    the plugins registered -- columns after the built-in ones, normalised with them, NaN policy by substring."""
    sys.path.insert(0, str(HERE.parent))
    import user_plugins as up

    out = {}
    # (1) the example's shape: 5 channels x 10 s of uniform noise at 1 kHz, 10 Hz features, default settings
    #     (notch + re-reference + z-score; sharp waves / bursts / fft / welch / hjorth ... enabled)
    nm.add_custom_feature("channel_mean", up.ChannelMean)
    try:
        rng = np.random.default_rng(2024)
        data = rng.random((5, 10000))
        s = nm.NMSettings.get_default()
        st = nm.Stream(sfreq=1000, data=data, settings=s, sampling_rate_features_hz=10, verbose=False)
        with tempfile.TemporaryDirectory() as td:
            df = st.run(out_dir=td, save_csv=False)
        out.update({"ex_data": data, "ex_settings_json": dump(st.settings), "ex_columns": np.array(list(df.columns)),
                    "ex_values": df.to_numpy(dtype=np.float64),
                    "ex_channels_json": json.dumps(st.channels.to_dict("list"))})
        print("user_features example", df.shape)
        # (2) two plugins (the second one is stateful and emits a "psd" key that the normaliser must skip), re-reference
        #     only, z-score on, a NaN stretch in one channel
        nm.add_custom_feature("hop_stats", up.HopStats)
        data = synth(4, 4000, 1000, seed=77)
        data[2, 2100:2130] = np.nan
        s = _small_settings()
        s.postprocessing.feature_normalization = True
        s.feature_normalization_settings.normalization_time_s = 1
        st, df = _run_stream(data, 1000, s)
        out.update({"two_data": data, "two_settings_json": dump(st.settings), "two_columns": np.array(list(df.columns)),
                    "two_values": df.to_numpy(dtype=np.float64),
                    "two_channels_json": json.dumps(st.channels.to_dict("list"))})
        print("user_features two plugins", df.shape)
        # (3) no pre-processing, no normaliser: the plugins see the float64 window itself
        s = _small_settings()
        s.preprocessing = []
        st, df = _run_stream(data, 1000, s)
        out.update({"raw_settings_json": dump(st.settings), "raw_columns": np.array(list(df.columns)),
                    "raw_values": df.to_numpy(dtype=np.float64)})
        print("user_features no preprocessing", df.shape)
    finally:
        for name in ("channel_mean", "hop_stats"):
            if name in nm.user_features:
                nm.remove_custom_feature(name)
    np.savez_compressed(HERE / "user_features.npz", **out)

def case_dc_offsets():
    """Recordings whose channels sit on DC offsets of 10^3 and 10^5 times their signal (electrode potentials, amplifier
    offsets): the reference is float64 end to end (stream/data_processor.py:238-260) and does not care -- an fp32 engine
    has to carry the constants next to the signal.  Feature classes on one window each, and the stream with the default
    common-average re-reference in front."""
    def mut(s):
        for f in ("activity", "mobility", "complexity"):
            setattr(s.bandpass_filter_settings.bandpower_features, f, True)
        s.fft_settings.features.max = True
        s.stft_settings.features.median = True
    for tag, ratio, seed in (("1e3", 1e3, 41), ("1e5", 1e5, 42)):
        x = synth(4, 1000, 1000, seed, dc=False)
        d = 50.0 * ratio * np.array([1.0, -1.0, 0.37, -0.61])
        feature_case(f"feat_dc{tag}", 1000, x + d[:, None], mut)
    out = {"sfreq": 1000}
    rng = np.random.default_rng(43)
    t = np.arange(6000) / 1000.0
    for tag, ratio in (("1e3", 1e3), ("1e5", 1e5)):
        sig = rng.standard_normal((5, 6000)) * 20 + 8 * np.sin(2 * np.pi * 21 * t) + 4 * np.sin(2 * np.pi * 9 * t + 1.0)
        data = sig + (20.0 * ratio * np.array([1.0, -0.8, 0.55, 0.9, -0.35]))[:, None]
        s = nm.NMSettings.get_default()
        s.features.bandpass_filter = True
        s.features.stft = True
        s.preprocessing = ["re_referencing"]
        s.postprocessing.feature_normalization = False
        s.sampling_rate_features_hz = 5
        st, df = _run_stream(data, 1000, s)
        out[f"{tag}_data"] = data
        out[f"{tag}_settings_json"] = dump(st.settings)
        out[f"{tag}_columns"] = np.array(list(df.columns))
        out[f"{tag}_values"] = df.to_numpy(dtype=np.float64)
        out[f"{tag}_channels_json"] = json.dumps(st.channels.to_dict("list"))
        print("dc pipeline", tag, df.shape)
    np.savez_compressed(HERE / "pipeline_dc_offsets.npz", **out)


def _read_brainvision(vhdr: Path):
    """Minimal BrainVision reader for the reference's own test recording (binary, multiplexed IEEE_FLOAT_32): ->
    (float32 [n_samples, n_channels] as stored, names, resolution * unit scale per channel, sfreq).  MNE's reader
    (`mne.io.read_raw_brainvision`, behind `nm.io.read_BIDS_data`, io.py) returns `stored * resolution * 1e-6` (micro-volt
    units -> volt) as float64; MNE itself is not installable in this image."""
    import configparser

    text = vhdr.read_text(encoding="utf-8")
    cp = configparser.ConfigParser(allow_no_value=True, delimiters=("=",), comment_prefixes=(";",), interpolation=None)
    cp.optionxform = str
    cp.read_string(text[text.index("[Common Infos]"):])
    assert cp["Common Infos"]["DataOrientation"] == "MULTIPLEXED" and cp["Binary Infos"]["BinaryFormat"] == "IEEE_FLOAT_32"
    n_ch = int(cp["Common Infos"]["NumberOfChannels"])
    sfreq = 1e6 / float(cp["Common Infos"]["SamplingInterval"])
    names, scale = [], []
    for i in range(n_ch):
        name, _, res, unit = cp["Channel Infos"][f"Ch{i + 1}"].split(",")[:4]
        assert unit == "\u00b5V"
        names.append(name)
        scale.append(float(res) * 1e-6)
    stored = np.fromfile(vhdr.parent / cp["Common Infos"]["DataFile"], dtype="<f4").reshape(-1, n_ch)
    return stored, names, np.array(scale), sfreq


def case_real_recording():
    """The recording the reference's own tests run on (tests/conftest.py:8-69: `nm.io.read_BIDS_data` of
    py_neuromodulation/data/sub-testsub/ses-EphysMedOff/ieeg/*_ieeg.vhdr -- 10 channels, 1 kHz, 19 s), the channel table
    its fixtures build (`nm.utils.set_channels(reference="default", ..., target_keywords=...)`, utils/channels.py:13-22:
    ECoG to the common average, the three LFP contacts bipolar, MOV_RIGHT the target), all nine hot-path feature
    families behind the default pre-processing (notch + re-reference), 10 Hz, with the default z-score and without."""
    import pandas as pd

    ieeg = Path(ref_shim.REFERENCE_ROOT) / "py_neuromodulation/data/sub-testsub/ses-EphysMedOff/ieeg"
    stored, names, scale, sfreq = _read_brainvision(next(ieeg.glob("*_ieeg.vhdr")))
    data = stored.T.astype(np.float64) * scale[:, None]
    tsv = pd.read_csv(next(ieeg.glob("*_channels.tsv")), sep="\t")
    assert tsv["name"].to_list() == names
    to_mne = {"DBS": "dbs", "ECOG": "ecog", "MISC": "misc", "SEEG": "seeg"}   # (raw.get_channel_types(): lower case)
    channels = nm.utils.set_channels(ch_names=names, ch_types=[to_mne[t] for t in tsv["type"]], reference="default",
                                     bads=[], new_names="default", used_types=("ecog", "dbs", "seeg"),
                                     target_keywords=("MOV_RIGHT",))
    out = {"sfreq": sfreq, "stored": stored, "scale": scale, "channels_json": json.dumps(channels.to_dict("list"))}
    for tag, norm in (("nonorm", False), ("zscore", True)):
        s = nm.NMSettings.get_default()
        s.features.enable_all()
        for f in ("fooof", "nolds", "coherence", "mne_connectivity", "bispectrum"):
            setattr(s.features, f, False)
        s.postprocessing.feature_normalization = norm
        s.sampling_rate_features_hz = 10
        st, df = _run_stream(data, sfreq, s, channels=channels, line_noise=50)
        out[f"{tag}_settings_json"] = dump(st.settings)
        out[f"{tag}_columns"] = np.array(list(df.columns))
        out[f"{tag}_values"] = df.to_numpy(dtype=np.float64)
        print("real_recording", tag, df.shape)
    np.savez_compressed(HERE / "real_recording.npz", **out)


def case_inf_members():
    """Several members of one re-reference group at +-inf in the same sample (amplifier rails, a decoder's overflow
    marker): nan_to_num makes them +-DBL_MAX (stream/data_processor.py:255) and `ref_matrix @ data`
    (processing/rereference.py:99-100) multiplies before it adds -- two members give the finite 2/(n-1) DBL_MAX on the
    other channels, opposite signs cancel, and only a sum beyond DBL_MAX is inf.  Default channel table (common average
    over 8 ECoG channels), re-reference only, 10 Hz."""
    rng = np.random.default_rng(61)
    t = np.arange(4000) / 1000.0
    data = rng.standard_normal((8, 4000)) * 40 + 12 * np.sin(2 * np.pi * 17 * t)
    data[[1, 3], 1250] = np.inf                      # two members
    data[[0, 4, 6], 1950] = np.inf                   # three members
    data[2, 2650], data[5, 2650] = np.inf, -np.inf   # both signs in one sample
    data[7, 3350], data[7, 3351] = -np.inf, -np.inf  # one member, two samples
    out = {"sfreq": 1000, "data": data}
    # "": the set the matrix-pipe spectrum kernel takes (FFT band means + time domain); "stft_": + STFT and Welch, the
    # wave-level kernel of the default window
    for tag, more in (("", ()), ("stft_", ("stft", "welch"))):
        s = nm.NMSettings.get_default()
        s.reset()
        s.features.fft = s.features.raw_hjorth = s.features.linelength = s.features.return_raw = True
        for f in more:
            setattr(s.features, f, True)
        s.preprocessing = ["re_referencing"]
        s.postprocessing.feature_normalization = False
        s.sampling_rate_features_hz = 10
        st, df = _run_stream(data, 1000, s)
        out.update({tag + "settings_json": dump(st.settings), tag + "columns": np.array(list(df.columns)),
                    tag + "values": df.to_numpy(dtype=np.float64), "channels_json": json.dumps(st.channels.to_dict("list"))})
        print("inf_members", tag, df.shape, "huge", int((np.abs(out[tag + "values"]) >= 1e37).sum()),
              "nan", int(np.isnan(out[tag + "values"]).sum()))
    np.savez_compressed(HERE / "inf_members.npz", **out)


def case_dc_nan():
    """NaN samples on a channel 10^5 spreads off zero, NO re-reference in front (the notch / the features read the rows
    directly): nan_to_num makes them the VALUE 0 (stream/data_processor.py:255) -- a step of the offset's size in that
    channel, which its stateful features (burst history and thresholds) keep seeing on later hops."""
    rng = np.random.default_rng(71)
    t = np.arange(7000) / 1000.0
    data = rng.standard_normal((3, 7000)) * 20 + 9 * np.sin(2 * np.pi * 19 * t) * (np.sin(2 * np.pi * 0.9 * t) > 0)
    data += (20.0 * 1e5 * np.array([1.0, -0.5, 0.3]))[:, None]
    data[1, 2500:2506] = np.nan
    out = {"sfreq": 1000, "data": data}
    for tag, pre in (("notch", ["notch_filter"]), ("nopre", [])):
        s = nm.NMSettings.get_default()
        s.reset()
        s.features.fft = s.features.raw_hjorth = s.features.linelength = s.features.return_raw = s.features.bursts = True
        s.bursts_settings.time_duration_s = 3
        s.preprocessing = pre
        s.postprocessing.feature_normalization = False
        s.sampling_rate_features_hz = 5
        import pandas as pd
        ch = pd.DataFrame({"name": ["a", "b", "c"], "rereference": ["None"] * 3, "used": [1] * 3, "target": [0] * 3,
                           "type": ["ecog"] * 3, "status": ["good"] * 3, "new_name": ["a", "b", "c"]})
        st, df = _run_stream(data, 1000, s, channels=ch)
        out[f"{tag}_settings_json"] = dump(st.settings)
        out[f"{tag}_columns"] = np.array(list(df.columns))
        out[f"{tag}_values"] = df.to_numpy(dtype=np.float64)
        out["channels_json"] = json.dumps(ch.to_dict("list"))
        print("dc_nan", tag, df.shape, "nan entries", int(np.isnan(out[f"{tag}_values"]).sum()))
    np.savez_compressed(HERE / "dc_nan.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # regenerate selected cases only: make_golden.py bandpower_kalman ...
        for name in sys.argv[1:]:
            globals()["case_" + name]()
        sys.exit(0)
    case_bandpower_kalman()
    case_preprocessing_filter()
    case_raw_normalizer()
    case_norm_methods()
    case_short_windows()
    case_schedule()
    case_feat_1k()
    case_feat_2k()
    case_special_rows()
    case_bursts_sequence()
    case_c5_degenerate()
    case_sharpwave_tests()
    case_long_windows()
    case_ragged_rawnorm()
    case_pipeline()
    case_nan_and_channels()
    case_notch()
    case_resample_quirk()
    case_output_files()
    case_user_features()
    case_ragged_bursts()
    case_dc_offsets()
    case_real_recording()
    case_inf_members()
    case_dc_nan()

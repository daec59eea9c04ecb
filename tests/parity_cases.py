"""Parity test bodies shared by tests/test_kernel_logic_emu.py (CPU, logic emulator) and
tests/test_gpu_parity.py (-m gpu, the real libnmx.so on the MI355X).  Each takes the loaded
library binding; see tests/parity.py for the tolerance policy."""

import numpy as np
import pytest

from tests import parity

FEATURE_CASES = ["feat_1k", "feat_1k_nolog", "feat_2k", "feat_special_rows"]


def case_feature_cases_match_reference_goldens(lib, case):
    n_bad, report, worst = parity.run_feature_case(lib, case)
    assert n_bad == 0, f"{case}: {n_bad} features outside tolerance\n{report}"


def case_dc_offsets(lib):
    """Channels on DC offsets of 10^3 and 10^5 times their signal (tests/golden/make_golden.py: case_dc_offsets, the
    reference's own float64 results): every feature class on one window, and the stream behind the default common-average
    re-reference -- at the STATED tolerances (1e-5 relative; 1e-5 absolute on log10-valued features), no verifier:
    the engine splits the constants off before anything is rounded to float32 and carries them next to the signal
    (nmx_engine_dc.inc)."""
    import json

    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    for case in ("feat_dc1e3", "feat_dc1e5"):
        # (at 10^5 the sharp-wave pre-filter's output IS the offset's edge transient, 10^4 times the signal: its float32
        # samples bound what any feature read from them can resolve -- 1e-5 of THAT series' amplitude)
        n_bad, report, worst = parity.run_feature_case(lib, case, forgive=False, sharpwave_series_amp=case == "feat_dc1e5")
        assert n_bad == 0, f"{case}: {n_bad} features outside tolerance\n{report}"
    g = load_golden("pipeline_dc_offsets")
    for tag in ("1e3", "1e5"):
        s = settings_from_json(g[f"{tag}_settings_json"])
        data = g[f"{tag}_data"]
        df = Stream(sfreq=1000.0, data=data, settings=s, line_noise=50, lib=lib).run(save_csv=False)
        cols = [str(c) for c in g[f"{tag}_columns"]]
        assert list(df.columns) == cols
        got, want = df.to_numpy(dtype=np.float64), g[f"{tag}_values"]
        assert got.shape == want.shape
        amp = float(np.abs(data - data.mean(axis=1, keepdims=True)).max())
        # sharp waves are read from a zero-padded pre-filter's output, which reaches their kernel as float32 samples: on a
        # window whose level the re-reference leaves at L that series is an edge transient of size ~L, and 1e-5 of ITS
        # amplitude is what "1e-5 of an amplitude" means for a feature read from it (run_feature_case above)
        from oracle import nm_oracle as orc
        ch = json.loads(str(g[f"{tag}_channels_json"]))
        st_, en_, _ = orc.window_schedule(data.shape[1], 1000.0, s.sampling_rate_features_hz, s.segment_length_features_ms)
        pv = parity.PipelineVerifiers(s, ch, 1000.0, data, st_, 1000, line_noise=50)
        fam = [parity.family_of(k) for k in cols]
        pick = lambda seq, name: [v for v, f in zip(seq, fam) if (f == name if name else f not in ("sharpwave", "bursts"))]   # noqa: E731
        for r in range(len(got)):
            n_bad, rep, _ = parity.compare(pick(cols, None), pick(got[r], None), pick(want[r], None), s, 1000.0, amp, 1000)
            assert n_bad == 0, f"offset / signal = {tag}, row {r}\n{rep}"
            level = float(np.abs(pv.window(r)).max())
            n_bad, rep, _ = parity.compare(pick(cols, "sharpwave"), pick(got[r], "sharpwave"), pick(want[r], "sharpwave"), s, 1000.0,
                                           level, 1000)
            assert n_bad == 0, f"offset / signal = {tag}, row {r} (sharp waves)\n{rep}"
            # bursts are DECISIONS (envelope sample >= a history quantile): the one family whose misses a conditioning
            # report may explain here -- a sample within fp32 rounding of its threshold
            n_bad, rep, _ = parity.compare(pick(cols, "bursts"), pick(got[r], "bursts"), pick(want[r], "bursts"), s, 1000.0, amp, 1000,
                                           verifier=pv.row(r))
            assert n_bad == 0, f"offset / signal = {tag}, row {r} (bursts)\n{rep}"


def case_pipelined_f64_batch(lib):
    """engine.process_batch_f64 with its two conversions on threads next to the copies and kernels (nmx_plan_set_pipeline:
    input published slice by slice, rows widened as the chunks land) == the plain batch, bit for bit, NaN mask included;
    with and without host offsets (a row far off zero: the split subtracts before the cast)."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.raw_hjorth = s.features.linelength = s.features.return_raw = True
    C, T = 12, 90000          # > 2^20 samples, 891 hops: several chunks
    rng = np.random.default_rng(5)
    for level in (0.0, 1e6):
        data = rng.standard_normal((C, T)) * 30 + rng.uniform(-100, 100, (C, 1)) + level
        data[3, 40000:40007] = np.nan
        starts = np.arange(0, T - 1000 + 1, 100)
        ch = [f"ch{i}" for i in range(C)]
        a = HotPathEngine(s, ch, 1000.0, lib=lib)
        want, wmask = a.process_batch(data, starts, want_nan_mask=True)
        a.close()
        b = HotPathEngine(s, ch, 1000.0, lib=lib)
        got, gmask = b.process_batch_f64(data, starts, want_nan_mask=True)
        assert (b._dc is not None and b._dc_any) == (level != 0.0)
        b.close()
        assert got.dtype == np.float64 and got.shape == want.shape
        np.testing.assert_array_equal(got, want.astype(np.float64))
        np.testing.assert_array_equal(gmask, wmask)
        assert wmask.any()


def case_multi_device_pipelined_batch(lib, devices=(0, 0, 0)):
    """MultiDeviceProcessor's pipelined batch (one thread stages every part slice by slice -- group sums + rows, or one
    shared cast --, every part widens into its columns of the table as its chunks land) == the same processor with the
    passes in a row (NMX_PIPELINE=0), bit for bit: local and replicated input, float32 and float64 recordings, NaN policy,
    the z-score normaliser inside the plans."""
    import os

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.sharding import MultiDeviceProcessor

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.raw_hjorth = s.features.fft = s.features.return_raw = True
    s.preprocessing = ["notch_filter", "re_referencing"]
    C, T = 7, 33000
    rng = np.random.default_rng(12)
    data = rng.standard_normal((C, T)) * 30 + rng.uniform(-100, 100, (C, 1))
    data[4, 20000:20005] = np.nan
    starts = np.arange(0, T - 1000 + 1, 100)
    channels = chmod.get_default_channels_from_data(data)
    for local in (True, False):
        for dtype in (np.float64, np.float32):
            x = data.astype(dtype)
            tables = []
            for pipe in ("1", "0"):
                dp = MultiDeviceProcessor(1000.0, s, channels, line_noise=50, devices=list(devices), local_input=local, lib=lib)
                dp.pipeline_min = (1, 0)
                assert dp.local_input == local
                os.environ["NMX_PIPELINE"] = pipe
                try:
                    dp.process_batch(-3 * x + 1000, starts)   # (another recording first: the staging arrays hold ITS samples
                    dp.reset()                                #  when the plans look at the first window of the next one)
                    tables.append(dp.process_batch(x, starts))
                    again = dp.process_batch(x[:, :5000], starts[:37])   # a second batch: state carried, counters re-armed
                finally:
                    del os.environ["NMX_PIPELINE"]
                    dp.close()
                tables.append(again)
            assert tables[0].dtype == np.float64 and tables[0].shape == (len(starts), len(dp.keys))
            assert np.isnan(tables[0]).any() and not np.isnan(tables[0]).all(axis=0).any()
            np.testing.assert_array_equal(tables[0], tables[2])
            np.testing.assert_array_equal(tables[1], tables[3])


def case_sharpwave_reference_test_inputs(lib):
    """Impulse / sine / plateau inputs of the reference's tests/test_sharpwave.py."""
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import golden_dict, load_golden, settings_from_json

    g = load_golden("sharpwave_tests")
    s = settings_from_json(g["settings_json"])
    ch = [str(c) for c in g["ch_names"]]
    eng = HotPathEngine(s, ch, 1000.0, lib=lib, features=["sharpwave_analysis"],
                        sharpwave_taps=[g["sw_taps_0"], g["sw_taps_1"]])
    out = eng.process_window(g["data"])
    want = golden_dict(g, "sharpwave")
    assert list(want) == eng.keys
    ver = parity.Verifier(s, ch, 1000.0, np.asarray(g["data"], np.float64), sw_taps=[g["sw_taps_0"], g["sw_taps_1"]])
    n_bad, report, _ = parity.compare(eng.keys, out, list(want.values()), s, 1000.0, 5.0, eng.W, verifier=ver)
    assert n_bad == 0, report


def case_bursts_sequence_state_across_batches(lib):
    """51 consecutive windows, ring of 2 s: the threshold state (top-K of the history) carries
    across calls and across the ring-overflow regime exactly like the reference's buffer."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("bursts_sequence")
    s = settings_from_json(g["settings_json"])
    ch = [str(c) for c in g["ch_names"]]
    sfreq = float(g["sfreq"])
    data = g["data"]
    starts, _, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz,
                                       s.segment_length_features_ms)
    keys = [str(k) for k in g["keys"]]
    want = g["values"]
    amp = float(np.abs(data).max())
    W = int(s.segment_length_features_ms / 1000 * sfreq)
    tracer = parity.BurstTracer(s, ch, sfreq, lambda i: data[:, starts[i]:starts[i] + W], taps=g["bursts_taps"])
    for split in (len(starts), 7):   # one batch, then several batches + single windows
        eng = HotPathEngine(s, ch, sfreq, lib=lib, features=["bursts"], bank_taps=None)
        assert eng.keys == keys
        rows = []
        i = 0
        while i < len(starts):
            n = min(split, len(starts) - i)
            if n == 1 or (split == 7 and i >= 28):
                rows.append(eng.process_window(data[:, starts[i]:starts[i] + eng.W])[None])
                n = 1
            else:
                rows.append(eng.process_batch(data, starts[i:i + n]))
            i += n
        got = np.concatenate(rows)
        n_bad = 0
        for r in range(len(starts)):
            ver = parity.Verifier(s, ch, sfreq, data[:, starts[r]:starts[r] + eng.W], bursts=lambda r=r: tracer.at(r))
            b, rep, _ = parity.compare(keys, got[r], want[r], s, sfreq, amp, eng.W, verifier=ver)
            n_bad += b
            assert b == 0, f"window {r}\n{rep}"
        # and tight agreement for the overwhelming majority of entries
        close = np.isclose(got, want, rtol=1e-4, atol=1e-6)
        assert close.mean() > 0.98
        eng.close()


def case_preprocessing_notch_and_reref(lib):
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings, fir_design
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import load_golden

    g = load_golden("notch_unpinned")
    s = NMSettings.get_default()
    for sfreq in (1000, 2000):
        x, y = g[f"x_{sfreq}"], g[f"y_{sfreq}"]
        s.segment_length_features_ms = 1000
        taps = fir_design.notch_bank(sfreq, 50)
        np.testing.assert_allclose(taps, g[f"taps_{sfreq}"], rtol=0, atol=1e-14)
        R = np.array([[1.0, -1.0], [-0.5, 1.0]])
        eng = HotPathEngine(s, ["a", "b"], float(sfreq), lib=lib, features=["raw_hjorth"],
                            notch_taps=taps, ref_matrix=R)
        got = eng.preprocess_window(x)
        want = orc.NotchFilter(sfreq, taps=taps).process(R @ x)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * np.abs(want).max())
        # notch only == reference glue output stored in the golden
        eng2 = HotPathEngine(s, ["a", "b"], float(sfreq), lib=lib, features=["raw_hjorth"], notch_taps=taps)
        np.testing.assert_allclose(eng2.preprocess_window(x), y, rtol=0, atol=2e-5 * np.abs(y).max())
        eng.close()
        eng2.close()


def case_filter_window_matches_mnefilter_shape_and_values(lib):
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("feat_1k")
    s = settings_from_json(g["settings_json"])
    ch = [str(c) for c in g["ch_names"]]
    eng = HotPathEngine(s, ch, 1000.0, lib=lib, features=["bandpass_filter"], bank_taps=g["bank_taps"])
    y = eng.filter_window(g["data"])
    assert y.shape == (4, 4, 1000)
    np.testing.assert_allclose(y[:2], g["bank_filtered"], rtol=0, atol=2e-5 * np.abs(g["bank_filtered"]).max())
    eng.close()


def case_nan_mask_and_clean_on_load(lib):
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 1000)) * 10
    x[1, 10:20] = np.nan
    eng = HotPathEngine(s, ["a", "b", "c"], 1000.0, lib=lib, features=["raw_hjorth", "linelength"])
    out, mask = eng.process_window(x, want_nan_mask=True)
    assert mask.tolist() == [False, True, False]
    want = {}
    xc = np.nan_to_num(x)
    want.update(orc.Hjorth(s, ["a", "b", "c"], 1000.0).calc_feature(xc))
    want.update(orc.LineLength(s, ["a", "b", "c"], 1000.0).calc_feature(xc))
    np.testing.assert_allclose(out, np.array(list(want.values())), rtol=1e-5)
    eng.close()


def _pipeline_case(lib, tag, rtol_norm=False):
    """README demo shape through the Stream mirror vs the reference DataFrame."""
    import json

    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("pipeline_readme")
    s = settings_from_json(g[f"{tag}_settings_json"])
    st = Stream(sfreq=float(g["sfreq"]), data=g["data"], settings=s, line_noise=50, lib=lib)
    df = st.run(save_csv=False)
    cols = [str(c) for c in g[f"{tag}_columns"]]
    assert list(df.columns) == cols, "DataFrame columns / order differ from the reference"
    want = g[f"{tag}_values"]
    got = df.to_numpy(dtype=np.float64)
    assert got.shape == want.shape
    return s, cols, got, want


def _pipeline_verifiers(tag, s):
    """Per-row verifiers for the README-shape goldens (oracle pre-processing of the same windows)."""
    import json

    from oracle import nm_oracle as orc
    from tests.helpers import load_golden

    g = load_golden("pipeline_readme")
    ch = json.loads(str(g[f"{tag}_channels_json"]))
    data, sfreq = g["data"], float(g["sfreq"])
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz,
                                          s.segment_length_features_ms)
    return parity.PipelineVerifiers(s, ch, sfreq, data, starts, int(ends[0] - starts[0]), line_noise=50, ends=ends)


def case_pipeline_readme_no_normalisation(lib):
    s, cols, got, want = _pipeline_case(lib, "reref_nonorm")
    pv = _pipeline_verifiers("reref_nonorm", s)
    for r in range(len(got)):
        n_bad, rep, _ = parity.compare(cols[:-1], got[r, :-1], want[r, :-1], s, 1000.0, 1.0, 1000,
                                       verifier=pv.row(r))
        assert n_bad == 0, f"row {r}\n{rep}"
    np.testing.assert_array_equal(got[:, -1], want[:, -1])  # time column


def case_pipeline_readme_default_zscore(lib):
    """notch + CAR + z-score normalisation (the default pipeline) and z-score without pre-processing.
    A z-score divides by the spread of the last 30 s, so it amplifies the fp32 rounding of a feature by
    value / std -- no fixed tolerance fits.  The composition is therefore verified stage by stage:
      (1) the un-normalised features of the same run against the reference golden, 1e-5 policy;
      (2) the engine's normalised output against the float64 oracle normaliser applied to the ENGINE'S
          OWN un-normalised rows (same fp32 inputs on both sides): 1e-5 relative + 2e-6;
      (3) first row un-normalised (normalization.py:93-97) and time column exact.
    The reference's normalised DataFrame is compared as a sanity check on top (median error)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden

    g = load_golden("pipeline_readme")
    for tag, raw_tag in (("default", "default_nonorm"), ("nopre_norm", None)):
        s, cols, got, want = _pipeline_case(lib, tag)
        np.testing.assert_array_equal(got[:, -1], want[:, -1])
        keys = cols[:-1]
        # the same run without the normaliser
        s_raw = type(s)(**s.to_dict())
        s_raw.postprocessing.feature_normalization = False
        raw = Stream(sfreq=float(g["sfreq"]), data=g["data"], settings=s_raw, line_noise=50, lib=lib).run(
            save_csv=False).to_numpy(dtype=np.float64)[:, :-1]
        pv = _pipeline_verifiers(tag, s_raw)
        if raw_tag is not None:   # (1) against the reference's un-normalised DataFrame
            assert [str(c) for c in g[f"{raw_tag}_columns"]] == cols
            ref_raw = g[f"{raw_tag}_values"]
            for r in range(len(raw)):
                n_bad, rep, _ = parity.compare(keys, raw[r], ref_raw[r, :-1], s_raw, 1000.0, 1.0, 1000,
                                               verifier=pv.row(r))
                assert n_bad == 0, f"{tag} raw row {r}\n{rep}"
        else:                     # (1) against the oracle (no reference golden without normalisation)
            import json

            rows = orc.run_stream(g["data"], float(g["sfreq"]), s_raw, json.loads(str(g[f"{tag}_channels_json"])))
            for r in range(len(raw)):
                n_bad, rep, _ = parity.compare(keys, raw[r], [rows[r][k] for k in keys], s_raw, 1000.0, 1.0, 1000,
                                               verifier=pv.row(r))
                assert n_bad == 0, f"{tag} raw row {r}\n{rep}"
        # (2) normaliser on the engine's own rows
        norm = orc.FeatureNormalizer(s)
        want_n = np.stack([norm.process(r.copy()) for r in raw])
        np.testing.assert_array_equal(got[0, :-1], raw[0])                      # (3)
        np.testing.assert_allclose(got[:, :-1], want_n, rtol=1e-5, atol=2e-6, err_msg=tag)
        # sanity against the reference's normalised values
        err = np.abs(got[1:, :-1] - want[1:, :-1])
        assert np.nanmedian(err) < 1e-4, tag


def case_pipeline_nan_and_channel_table(lib):
    import json

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("pipeline_nan_channels")
    s = settings_from_json(g["settings_json"])
    for tag in ("nan", "mix"):
        ch = json.loads(str(g[f"{tag}_channels_json"]))
        st = Stream(sfreq=1000.0, channels=ch, settings=s, line_noise=50, lib=lib)
        df = st.run(g[f"{tag}_data"], save_csv=False)
        cols = [str(c) for c in g[f"{tag}_columns"]]
        assert list(df.columns) == cols
        got, want = df.to_numpy(dtype=np.float64), g[f"{tag}_values"]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        data = g[f"{tag}_data"]
        starts, ends, _ = orc.window_schedule(data.shape[1], 1000.0, s.sampling_rate_features_hz,
                                              s.segment_length_features_ms)
        pv = parity.PipelineVerifiers(s, ch, 1000.0, data, starts, 1000, line_noise=50)
        for r in range(len(got)):
            n_bad, rep, _ = parity.compare(cols, got[r], want[r], s, 1000.0, 30.0, 1000, verifier=pv.row(r))
            assert n_bad == 0, f"{tag} row {r}\n{rep}"


def case_bursts_steady_state_vs_oracle(lib):
    """Ring of 5 s (K = 1251 > fringe capacity): 40 fill hops, then 110 hops in the steady
    regime (fringe / pending-list algorithm of the threshold kernel), vs the CPU oracle; the same
    sequence is also fed in uneven batches and single windows (flush on batch boundaries)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.bursts_settings.time_duration_s = 5
    s.bursts_settings.frequency_bands = ["low_beta", "high_beta"]
    s = s.validate()
    sfreq, C, n_hops = 1000.0, 2, 150
    T = 1000 + (n_hops - 1) * 100
    rng = np.random.default_rng(21)
    t = np.arange(T) / sfreq
    amp = 1 + 0.8 * np.sin(2 * np.pi * 0.3 * t) + t / t[-1]          # non-stationary power
    data = rng.standard_normal((C, T)) * 20 + 30 * amp * np.sin(2 * np.pi * 18 * t)
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * 100
    ob = orc.Bursts(s, ch, sfreq)
    want, traces = [], []
    for a in starts:
        d = ob.calc_feature(data[:, a:a + 1000])
        want.append([float(v) for v in d.values()])
        traces.append(parity.BurstTrace(ob))
    want = np.array(want)
    keys = list(d.keys())
    amp_scale = float(np.abs(data).max())
    for plan in ([n_hops], [37, 1, 1, 50, 13, 48]):
        eng = HotPathEngine(s, ch, sfreq, lib=lib, features=["bursts"], bank_taps=None)
        assert eng.keys == keys
        rows, i = [], 0
        for n in plan:
            if n == 1:
                rows.append(eng.process_window(data[:, starts[i]:starts[i] + 1000])[None])
            else:
                rows.append(eng.process_batch(data, starts[i:i + n]))
            i += n
        got = np.concatenate(rows)
        for r in range(n_hops):
            ver = parity.Verifier(s, ch, sfreq, data[:, starts[r]:starts[r] + 1000], bursts=traces[r])
            b, rep, _ = parity.compare(keys, got[r], want[r], s, sfreq, amp_scale, 1000, verifier=ver)
            assert b == 0, f"plan {plan} hop {r}\n{rep}"
        assert np.isclose(got, want, rtol=1e-4, atol=1e-6).mean() > 0.97
        eng.close()


# ---- odd sizes / generic radices ----------------------------------------------------------------

def case_ragged_float_sfreq_stream(lib):
    """tests/test_feature_sampling_rates.py shape: sfreq = 1111.111 Hz makes the generator cut windows
    of 1111 and 1112 samples (= 11 * 101 -> generic prime radices in the FFT); Stream handles the
    ragged schedule with one plan per length.  Compared with the oracle hop by hop."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = s.features.return_raw = True
    s.preprocessing = ["re_referencing"]
    s.postprocessing.feature_normalization = False
    sfreq = 1111.111
    rng = np.random.default_rng(31)
    data = rng.standard_normal((3, 6000)) * 10 + rng.uniform(-50, 50, (3, 1))
    df = Stream(sfreq, data=data, settings=s, line_noise=50, lib=lib).run(save_csv=False)
    ch = chmod.get_default_channels_from_data(data).to_dict("list")
    rows = orc.run_stream(data, sfreq, s, ch)
    assert list(df.columns) == list(rows[0].keys())
    assert len(df) == len(rows) and len({len(r) for r in rows}) == 1
    got = df.to_numpy(float)
    lens = set()
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, 1111, line_noise=50, ends=ends)
    for i, r in enumerate(rows):
        want = np.array(list(r.values()))
        n_bad, rep, _ = parity.compare(list(df.columns)[:-1], got[i, :-1], want[:-1], s, sfreq, 40.0, 1111,
                                       verifier=pv.row(i))
        assert n_bad == 0, f"hop {i}\n{rep}"
        assert got[i, -1] == want[-1]   # time column (test_timing.py)
    return lens


def case_odd_windows_and_spectra(lib):
    """STFT with an odd nperseg (333 = 9 * 37: full complex transform + generic radix 37), Welch
    averaging 3 segments (W = 2 * sfreq), FFT with return_spectrum, all four estimators."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.segment_length_features_ms = 2000
    s.features.disable_all()
    s.features.fft = s.features.welch = s.features.stft = True
    s.stft_settings.windowlength_ms = 333
    s.fft_settings.windowlength_ms = 1000
    s.fft_settings.return_spectrum = True
    for name in ("fft_settings", "welch_settings", "stft_settings"):
        for e in ("mean", "median", "std", "max"):
            setattr(s[name].features, e, True)
    s = s.validate()
    rng = np.random.default_rng(32)
    x = rng.standard_normal((3, 2000)) * 5 + 100 + np.sin(2 * np.pi * 20 * np.arange(2000) / 1000.0)
    ch = ["a", "b", "c"]
    eng = HotPathEngine(s, ch, 1000.0, lib=lib)
    got = eng.process_window(x)
    want = {}
    for cls in (orc.STFT, orc.FFT, orc.Welch):
        want.update(cls(s, ch, 1000.0).calc_feature(x))
    assert list(want) == eng.keys
    n_bad, rep, _ = parity.compare(eng.keys, got, list(want.values()), s, 1000.0, 20.0, 2000,
                                   verifier=parity.Verifier(s, ch, 1000.0, x))
    assert n_bad == 0, rep
    eng.close()


def case_reference_property_tests(lib):
    """The reference's own property tests restated on the engine (tests/test_osc_features.py,
    test_notch_filter.py, test_sharpwave.py, test_all_features.py)."""
    from py_neuromodulation_amd import NMSettings, features
    from py_neuromodulation_amd.processing import NotchFilter

    sfreq = 1000.0
    t = np.arange(1000) / sfreq
    np.random.seed(0)
    s = NMSettings.get_default()
    s.features.bandpass_filter = s.features.stft = True
    ch = ["ch1"]
    sine = (10 * np.sin(2 * np.pi * 16.5 * t) + 0.5 * np.random.random(1000))[None]  # off-bin tone in low_beta
    for cls, tag in ((features.FFT, "fft"), (features.Welch, "welch"), (features.STFT, "stft")):
        d = cls(s, ch, sfreq).calc_feature(sine)
        beta = max(d[f"ch1_{tag}_low_beta_mean"], d[f"ch1_{tag}_high_beta_mean"])
        assert beta > d[f"ch1_{tag}_theta_mean"] and beta > d[f"ch1_{tag}_alpha_mean"]
    d = features.BandPower(s, ch, sfreq).calc_feature(sine)
    assert max(d["ch1_bandpass_activity_low_beta"], d["ch1_bandpass_activity_high_beta"]) > \
        d["ch1_bandpass_activity_theta"]
    # constant input: non-DC FFT magnitude ~ 0 (log off)
    s2 = NMSettings.get_default()
    s2.fft_settings.log_transform = False
    d = features.FFT(s2, ch, sfreq).calc_feature(np.ones((1, 1000)))
    assert all(abs(v) < 1e-3 for v in d.values())
    # zeros / NaN input must not raise (tests/test_all_features.py)
    full = features.HotPathFeatures(s, ch, sfreq)
    full.calc_feature(np.zeros((1, 1000)))
    full.calc_feature(np.full((1, 1000), np.nan))
    # notch reduces power at the line frequency (tests/test_notch_filter.py)
    for fs in (500.0, 1000.0):
        tt = np.arange(int(fs)) / fs
        x = (np.sin(2 * np.pi * 50 * tt) + 0.1 * np.random.random(int(fs)))[None]
        y = NotchFilter(fs, line_noise=50).process(x)
        k = int(round(50 * len(tt) / fs))
        assert np.abs(np.fft.rfft(y[0]))[k] < 0.2 * np.abs(np.fft.rfft(x[0]))[k]
    # sharp waves: prominence grows with the impulse height (tests/test_sharpwave.py:65-93)
    prom = []
    for h in (1, 2, 3, 4):
        x = np.zeros((1, 1000))
        x[0, 100::200] = h
        d = features.SharpwaveAnalyzer(s, ch, sfreq).calc_feature(x)
        prom.append(d["ch1_Sharpwave_Max_prominence_range_5_80"])
    assert all(b > a for a, b in zip(prom, prom[1:]))
    # settings validation errors surface as ValueError / AssertionError like the reference's
    import pytest

    bad = NMSettings.get_default()
    bad.fft_settings.windowlength_ms = 2000   # longer than the segment
    with pytest.raises(AssertionError):
        features.FFT(bad, ch, sfreq)
    with pytest.raises(ValueError):
        NMSettings(fft_settings={"log_transform": "yes"})


def case_feature_normalizer_batches(lib):
    """nmx_norm_* (batch scan) == the reference's hop-by-hop Normalizer for "zscore", "mean", "median",
    "zscore-median" and the scikit-learn based "robust", "minmax", "quantile" (fitted on nan_to_num(history)):
    history carried across batches and through export/import, N - 1 trimming, NaN-aware statistics,
    constant columns (std 0 -> 1), the untouched first row, clip, and the "psd" column mask.
    Tolerance: statistics are float64 on both sides, values are fp32 -> 1e-5 rel / 2e-6 abs."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.processing import DeviceFeatureNormalizer

    rng = np.random.default_rng(7)
    n, F = 400, 37
    rows = (rng.standard_normal((n, F)) * rng.uniform(0.01, 30, F) + rng.uniform(-50, 50, F)).astype(np.float32)
    rows[:, 3] = 2.5                                   # constant column
    rows[:, 4] = np.cumsum(np.abs(rows[:, 4])) + 1e4   # drifting, mean >> std
    rows[rng.integers(0, n, 40), rng.integers(5, 12, 40)] = np.nan
    rows[100:180, 12] = np.nan                         # a whole history window of NaNs (N = 50)
    rows[60:75, 13] = -np.inf                          # log10 of a zero power (flat channel), ADVICE r1
    rows[200, 14] = np.inf
    rows[230:233, 14] = -np.inf                        # both signs inside one window

    def huge(a):   # nan_to_num(+-inf) is the dtype's max: float64 in the reference, fp32 on the device
        a = np.array(a, dtype=np.float64)
        a[np.abs(a) >= 1e37] = np.sign(a[np.abs(a) >= 1e37]) * np.inf
        return a
    mask = np.ones(F, dtype=np.uint8)
    mask[20:24] = 0
    rows[:, 15] = np.round(rows[:, 15])                # ties: repeated quantiles
    col16 = rows[:, 16].copy()
    rows[:, 17] = np.abs(rows[:, 17]) * 1e-6           # tiny values ...
    rows[::97, 17] = 40.0                              # ... and a huge one passing through the history every 97 hops
    for method, clip in (("zscore", 3), ("mean", 3), ("zscore", 0), ("median", 3), ("zscore-median", 3), ("median", 0),
                         ("robust", 3), ("robust", 0), ("minmax", 3), ("quantile", 3), ("quantile", 0)):
        # a nan_to_num'ed -inf feature (band power of a flat channel).  Not for the scikit-learn methods: np.nanmedian
        # of a 2-D history with fewer than 600 cells averages (low + high) also for odd counts, which overflows to -inf
        # for +-DBL_MAX; the kernel follows the large-array branch (np.median per column)
        rows[:, 16] = col16 if method in ("robust", "minmax", "quantile") else np.finfo(np.float32).min
        s = NMSettings.get_default()
        s.sampling_rate_features_hz = 10
        s.feature_normalization_settings.normalization_time_s = 5
        s.feature_normalization_settings.normalization_method = method
        s.feature_normalization_settings.clip = clip
        ref = orc.FeatureNormalizer(s)
        # (+-FLT_MAX is the fp32 image of the reference's nan_to_num'ed +-inf = +-DBL_MAX: the oracle sees the latter, where
        # e.g. the median of an even number of them overflows to -inf exactly as numpy's does in the reference)
        want = np.stack([ref.process(parity.widen_huge(r)) for r in rows[:, mask == 1]])
        dn = DeviceFeatureNormalizer(s, F, colmask=mask, lib=lib)
        got = [dn.process_batch(rows[:1]), dn.process_batch(rows[1:130])]
        state = dn.export_state()
        dn2 = DeviceFeatureNormalizer(s, F, colmask=mask, lib=lib)
        dn2.import_state(state)
        got.append(dn2.process_batch(rows[130:131]))
        got.append(dn2.process(rows[131])[None])       # the reference's one-vector call shape
        got.append(dn2.process_batch(rows[132:]))
        got = np.concatenate(got).astype(np.float64)
        np.testing.assert_array_equal(got[:, mask == 0], rows[:, mask == 0].astype(np.float64))
        np.testing.assert_allclose(huge(got[:, mask == 1]), huge(want), rtol=1e-5, atol=2e-6, err_msg=f"{method} clip={clip}")
        dn2.reset()
        np.testing.assert_array_equal(dn2.process_batch(rows[:1]), rows[:1])


def case_feature_normalizer_power(lib):
    """feature_normalization_method "power" (scikit-learn's PowerTransformer fitted on the history every hop,
    nmx_k_power.h): (1) against the REFERENCE's own FeatureNormalizer + scikit-learn (golden norm_methods.npz: 160
    hops, N = 50, NaN cells, a constant column, ties, a skewed positive column), with and without clip; (2) random
    rows in several batches with the history carried through export / import and the "psd" column mask, against the
    float64 oracle.  Tolerance 1e-5 relative + 2e-6 absolute (float64 fit on both sides, fp32 values) -- except on
    histories of fewer than 16 rows, where the Yeo-Johnson likelihood is so flat that lambda follows the rounding of
    the transcendental functions (oracle vs scipy differ by 5e-7 there themselves): 2e-3."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.processing import DeviceFeatureNormalizer
    from tests.helpers import load_golden

    def check(got, want, what, raw, n_hist, clip=0):
        """1e-5 relative + 2e-6 absolute; a miss is accepted only when the float64 oracle shows that lambdas whose
        likelihood is within float64 evaluation noise of the maximum move THAT output by as much
        (oracle.yeo_johnson_conditioning: near-constant-relative-spread columns and very short histories have a flat
        likelihood -- the reference's own answer is then one of many, decided by libm rounding)."""
        accepted = 0
        for i in range(len(want)):
            bad = np.where(~(np.abs(got[i] - want[i]) <= 1e-5 * np.abs(want[i]) + 2e-6))[0]
            for j in bad:
                if np.isnan(got[i][j]) and np.isnan(want[i][j]):
                    continue
                hist = np.nan_to_num(raw[max(0, i - n_hist + 1):i + 1, j].astype(np.float64))
                slack = orc.yeo_johnson_conditioning(hist, float(raw[i, j]))
                err = abs(got[i][j] - want[i][j])
                assert err <= 4 * slack + 1e-5 * abs(want[i][j]) + 2e-6, (
                    f"{what} hop {i} col {j}: got {got[i][j]!r}, want {want[i][j]!r}, conditioning slack {slack:.3g}")
                accepted += 1
        parity.note_forgiven("power_lambda", accepted, got.size)

    g = load_golden("norm_methods")
    rows = g["rows"].astype(np.float32)
    for clip in (3, 0):
        s = NMSettings.get_default()
        s.sampling_rate_features_hz = 10
        s.feature_normalization_settings.normalization_time_s = 5
        s.feature_normalization_settings.normalization_method = "power"
        s.feature_normalization_settings.clip = clip
        ref = orc.FeatureNormalizer(s)   # (the golden was computed from the float64 rows: the oracle sees the fp32 ones)
        want = np.stack([ref.process(r.astype(np.float64)) for r in rows])
        # the float64 golden itself, where fp32 input rounding cannot matter more than the tolerance
        dn = DeviceFeatureNormalizer(s, rows.shape[1], lib=lib)
        got = np.concatenate([dn.process_batch(rows[:1]), dn.process_batch(rows[1:70]), dn.process_batch(rows[70:])])
        check(got, want, f"golden clip={clip}", rows, 50)
    rng = np.random.default_rng(21)
    n, F = 120, 21
    rows = (rng.standard_normal((n, F)) * rng.uniform(0.01, 30, F) + rng.uniform(-50, 50, F)).astype(np.float32)
    rows[:, 3] = 2.5
    rows[:, 4] = np.log10(np.abs(rows[:, 4]) + 1e-4)
    rows[rng.integers(0, n, 15), rng.integers(5, 9, 15)] = np.nan
    mask = np.ones(F, dtype=np.uint8)
    mask[10:13] = 0
    s = NMSettings.get_default()
    s.sampling_rate_features_hz = 10
    s.feature_normalization_settings.normalization_time_s = 3
    s.feature_normalization_settings.normalization_method = "power"
    ref = orc.FeatureNormalizer(s)
    want = np.stack([ref.process(r.astype(np.float64)) for r in rows[:, mask == 1]])
    dn = DeviceFeatureNormalizer(s, F, colmask=mask, lib=lib)
    got = [dn.process_batch(rows[:1]), dn.process_batch(rows[1:40])]
    dn2 = DeviceFeatureNormalizer(s, F, colmask=mask, lib=lib)
    dn2.import_state(dn.export_state())
    got += [dn2.process_batch(rows[40:41]), dn2.process(rows[41])[None], dn2.process_batch(rows[42:])]
    got = np.concatenate(got).astype(np.float64)
    np.testing.assert_array_equal(got[:, mask == 0], rows[:, mask == 0].astype(np.float64))
    check(got[:, mask == 1], want, "random", rows[:, mask == 1], 30)


def case_stream_output_files(lib, tmp_path):
    """Stream.run leaves the files the REFERENCE's own Stream.run leaves (golden output_files.npz, written by
    the unmodified MsgPackFileWriter / _save_after_stream: utils/file_writer.py:53-118, stream/stream.py:426-453):
    same names, the same per-interval msgpack layout (rows per file, key order, float values), the same CSV
    header / line count / values, identical sidecar JSON and channels table text, the same settings keys;
    the default call deletes the per-interval files like the reference."""
    import json

    import msgpack
    import pandas as pd
    import yaml

    from py_neuromodulation_amd.file_writer import MsgPackFileWriter
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("output_files")
    s = settings_from_json(g["settings_json"])
    data = g["data"]
    st = Stream(sfreq=1000.0, data=data, settings=s, sampling_rate_features_hz=10, lib=lib)
    df = st.run(data, out_dir=tmp_path, experiment_name="sub7", save_csv=True, save_interval=10,
                delete_ind_batch_files_after_stream=False)
    out = tmp_path / "sub7"
    assert sorted(p.name for p in out.iterdir()) == [str(n) for n in g["file_names"]]
    # what the reference's run leaves on the object (stream/stream.py:213-219): its examples go on from there
    # (examples/plot_0_first_demo.py: nm.FeatureReader(feature_dir=stream.out_dir, feature_file=stream.experiment_name))
    assert (st.out_dir, st.experiment_name, st.save_csv, st.save_interval, st.return_df) == (tmp_path, "sub7", True, 10, True)
    assert st.batch_count == len(df) and st.is_stream_lsl is False
    cols = [str(c) for c in g["df_columns"]]
    assert list(df.columns) == cols and [str(t) for t in df.dtypes] == [str(t) for t in g["df_dtypes"]]
    want = g["df_values"]
    got = df.to_numpy(dtype=np.float64)
    for r in range(len(got)):
        n_bad, rep, _ = parity.compare(cols[:-1], got[r, :-1], want[r, :-1], s, 1000.0, 4.0, 1000,
                                       verifier=parity.Verifier(s, [c[:-len("_RawHjorth_Activity")] for c in cols[0:9:3]], 1000.0,
                                                                lambda r=r: _car3(data[:, r * 100:r * 100 + 1000]), raw=data[:, r * 100:r * 100 + 1000]))
        assert n_bad == 0, f"row {r}\n{rep}"
    np.testing.assert_array_equal(got[:, -1], want[:, -1])
    packs = sorted(out.glob("sub7-*.msgpack"), key=lambda p: int(p.stem.split("-")[1]))
    rows_per_file = []
    for i, p in enumerate(packs):
        with open(p, "rb") as f:
            d = msgpack.unpack(f)
        rows_per_file.append(len(d))
        if i == 0:
            assert list(d[0].keys()) == [str(k) for k in g["msgpack_first_keys"]]
            assert [type(v).__name__ for v in d[0].values()] == [str(t) for t in g["msgpack_first_types"]]
    assert rows_per_file == g["msgpack_rows_per_file"].tolist()
    csv_text = (out / "sub7_FEATURES.csv").read_text()
    assert csv_text.splitlines()[0] == str(g["csv_header"]) and len(csv_text.splitlines()) == int(g["csv_n_lines"])
    np.testing.assert_allclose(pd.read_csv(out / "sub7_FEATURES.csv").to_numpy(), got, rtol=1e-12, equal_nan=True)
    assert json.loads((out / "sub7_SIDECAR.json").read_text()) == json.loads(str(g["sidecar_json"]))
    assert (out / "sub7_SIDECAR.json").read_text() == str(g["sidecar_json"])
    assert (out / "sub7_channels.csv").read_text() == str(g["channels_csv"])
    assert list(yaml.safe_load((out / "sub7_SETTINGS.yaml").read_text()).keys()) == [str(k) for k in g["settings_yaml_keys"]]
    # the reference's default call: per-interval files are removed after the run
    st2 = Stream(sfreq=1000.0, data=data, settings=s, sampling_rate_features_hz=10, lib=lib)
    st2.run(data, out_dir=tmp_path / "d", experiment_name="sub7")
    assert sorted(p.name for p in (tmp_path / "d" / "sub7").iterdir()) == [str(n) for n in g["file_names_default_call"]]
    # reading the files back / the hop-by-hop interface of the writer (the reference's call shape)
    w = MsgPackFileWriter(name="sub7", out_dir=tmp_path)
    w.idx = len(packs)
    back = w.load_all()
    assert list(back.columns) == cols
    np.testing.assert_array_equal(back.to_numpy(), got)
    w2 = MsgPackFileWriter(name="live", out_dir=tmp_path)
    for i in range(3):
        w2.insert_data({"a": i, "b": None})
    w2.save()
    w2.save_as_csv(save_all_combined=True)
    assert pd.read_csv(tmp_path / "live" / "live_FEATURES.csv")["b"].tolist() == [0, 0, 0]
    w2.delete_ind_files()
    assert not list((tmp_path / "live").glob("*.msgpack"))


def _car3(w):
    """common average of three channels (the default channel table of a 3-row array)"""
    R = np.full((3, 3), -0.5)
    np.fill_diagonal(R, 1.0)
    return R @ np.asarray(w, np.float64)


def case_bandpower_kalman_sequence(lib):
    """41 consecutive hops of BandPower with kalman_filter (golden from the reference): the per
    (channel, band) filter state carries across batches, single-window calls and export/import.
    fp32 log-activity goes into a float64 filter: tolerance 1e-5 relative + 2e-6 absolute."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("bandpower_kalman")
    s = settings_from_json(g["settings_json"])
    ch = [str(c) for c in g["ch_names"]]
    sfreq = float(g["sfreq"])
    data = g["data"]
    starts, _, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz,
                                       s.segment_length_features_ms)
    keys = [str(k) for k in g["keys"]]
    want = g["values"]
    eng = HotPathEngine(s, ch, sfreq, lib=lib, features=["bandpass_filter"])
    assert eng.keys == keys
    rows = [eng.process_batch(data, starts[:17])]
    rows += [eng.process_window(data[:, a:a + eng.W])[None] for a in starts[17:20]]
    state = eng.export_state()
    eng2 = HotPathEngine(s, ch, sfreq, lib=lib, features=["bandpass_filter"])
    eng2.import_state(state)
    rows.append(eng2.process_batch(data, starts[20:]))
    got = np.concatenate(rows)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)
    # reset = fresh filters (x = [0, 1], P = cov([[1, 0], [0, 1]]))
    eng2.reset_state()
    np.testing.assert_allclose(eng2.process_batch(data, starts[:3]), want[:3], rtol=1e-5, atol=2e-6)
    eng.close()
    eng2.close()


def case_resampler(lib):
    """The plan's resampler stage == the restated mne.filter.resample (oracle/mne_restated.py, PARITY UNPINNED
    against MNE itself): down- and up-sampling, power-of-two and composite lengths, an odd resampled
    length (full complex inverse), NaN cleaning, and the engine path notch -> resample -> features.
    Tolerance: fp32 transforms of ~1e3..4e3 points on data of amplitude A: 2e-5 * A absolute."""
    from oracle import mne_restated as mr
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings, fir_design
    from py_neuromodulation_amd.engine import HotPathEngine
    from py_neuromodulation_amd.processing import Resampler

    rng = np.random.default_rng(11)
    for sf_old, sf_new, W in ((2000, 1000, 2000), (1000, 2000, 500), (2048, 1000, 2048), (4000, 1000, 4000),
                              (1000, 250, 1000), (1375, 500, 1375), (1000, 1000, 300)):
        t = np.arange(W) / sf_old
        x = rng.standard_normal((3, W)) * 20 + 50 * np.sin(2 * np.pi * 11 * t) + rng.uniform(-300, 300, (3, 1))
        if sf_old == sf_new:
            got = Resampler(sf_old, sf_new).process(x)     # (identity: the array itself, like the reference)
            assert got is x
        else:   # the plan's fp32 LDS resampler on one window (the stand-alone class: case_standalone_resampler_float64)
            got = HotPathEngine(
                NMSettings.get_default(), [f"c{i}" for i in range(3)], sf_new, features=["return_raw"],
                resample_from=sf_old, raw_window=W, window=int(round(sf_new / sf_old * W)), lib=lib).preprocess_window(x)
        want = mr.resample(x, up=sf_new / sf_old, down=1.0)
        assert got.shape == want.shape == (3, int(round(sf_new / sf_old * W)))
        amp = np.abs(x).max()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * amp, err_msg=f"{sf_old}->{sf_new}")
    # engine path: 2 kHz windows, notch at the raw rate, resample to 1 kHz, features at 1 kHz
    s = NMSettings.get_default()
    for f in s.features.get_enabled():
        setattr(s.features, f, False)
    s.features.fft = True
    s.features.raw_hjorth = True
    s.features.return_raw = True
    s.postprocessing.feature_normalization = False
    C, W = 2, 2000
    T = W + 5 * 200
    t = np.arange(T) / 2000
    x = rng.standard_normal((C, T)) * 20 + 30 * np.sin(2 * np.pi * 50 * t) + 25 * np.sin(2 * np.pi * 17 * t)
    x[0, 1200] = np.nan
    notch = fir_design.notch_bank(2000.0, 50)
    eng = HotPathEngine(s, ["a", "b"], 1000.0, resample_from=2000.0, notch_taps=notch, lib=lib)
    assert (eng.W_in, eng.W) == (2000, 1000)
    starts = np.arange(6) * 200
    got, mask = eng.process_batch(x, starts, want_nan_mask=True)
    assert mask[:, 0].tolist() == [a <= 1200 < a + W for a in starts] and not mask[:, 1].any()
    feats = [orc.Hjorth(s, ["a", "b"], 1000.0), orc.Raw(s, ["a", "b"], 1000.0), orc.FFT(s, ["a", "b"], 1000.0)]
    nf = orc.NotchFilter(2000.0, 50, taps=notch)
    for i, a in enumerate(starts):
        w = np.nan_to_num(x[:, a:a + W])
        y = mr.resample(nf.process(w), up=0.5, down=1.0)
        want = {}
        for f in feats:
            want.update(f.calc_feature(y))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], [want[k] for k in eng.keys], s, 1000.0,
                                       float(np.abs(w).max()), 1000, verifier=parity.Verifier(s, ["a", "b"], 1000.0, y, raw=w))
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()
    # Stream level: the consistent pipeline (features designed for the new rate) is opt-in; the default
    # reproduces the reference's raw-rate design (case_raw_resampling_reference_quirk)
    from py_neuromodulation_amd.stream import Stream

    s.preprocessing = ["raw_resampling", "notch_filter", "re_referencing"]
    xs = np.nan_to_num(x)
    st = Stream(sfreq=2000.0, data=xs, settings=s, lib=lib, resample_features_at_new_rate=True)
    df = st.run(xs, save_csv=False)
    assert len(df) == 6 and st.data_processor.sfreq_raw == 1000.0
    R = np.array([[1.0, -1.0], [-1.0, 1.0]])      # default channel table: common average of 2 channels
    names = [k for k in df.columns if k != "time"]
    for i, a in enumerate(starts):
        y = R @ mr.resample(nf.process(xs[:, a:a + W]), up=0.5, down=1.0)
        want = {}
        for f in (orc.Hjorth(s, ["ch0_avgref", "ch1_avgref"], 1000.0), orc.Raw(s, ["ch0_avgref", "ch1_avgref"], 1000.0),
                  orc.FFT(s, ["ch0_avgref", "ch1_avgref"], 1000.0)):
            want.update(f.calc_feature(y))
        n_bad, rep, _ = parity.compare(names, df.iloc[i][names].to_numpy(dtype=np.float64),
                                       [want[k] for k in names], s, 1000.0, float(np.abs(xs).max()), 1000,
                                       verifier=parity.Verifier(s, ["ch0_avgref", "ch1_avgref"], 1000.0, y, raw=xs[:, a:a + W]))
        assert n_bad == 0, f"stream hop {i}\n{rep}"


def case_preprocessing_filter(lib):
    """PreprocessingFilter.process vs the reference golden (chained zero-padded FIRs, default four
    stages incl. 1651-tap auto-length ones, and a two-stage subset) and inside the pipeline in front
    of the notch.  Tolerance: fp32 FFT convolution of amplitude-A data, 2e-5 * A absolute per stage."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings, fir_design
    from py_neuromodulation_amd.engine import HotPathEngine
    from py_neuromodulation_amd.processing import PreprocessingFilter
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("preprocessing_filter")
    sfreq = float(g["sfreq"])
    for tag in ("all", "two"):
        s = settings_from_json(g[f"{tag}_settings_json"])
        x, want = g[f"{tag}_x"], g[f"{tag}_y"]
        pf = PreprocessingFilter(s, sfreq)
        assert len(pf.taps) == int(g[f"{tag}_n_filters"])
        pf._engines[x.shape] = HotPathEngine(NMSettings.get_default(), [f"c{i}" for i in range(x.shape[0])], sfreq,
                                             features=["return_raw"], pre_taps=pf.taps, window=x.shape[1], lib=lib)
        got = pf.process(x)
        amp = np.abs(x).max()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * amp * len(pf.taps), err_msg=tag)
    # in the pipeline: preprocessing_filter -> notch -> features (batch of hops)
    s = NMSettings.get_default()
    for f in s.features.get_enabled():
        setattr(s.features, f, False)
    s.features.raw_hjorth = True
    s.features.linelength = True
    s.postprocessing.feature_normalization = False
    rng = np.random.default_rng(5)
    C, W, T = 2, 1000, 1500
    t = np.arange(T) / sfreq
    x = rng.standard_normal((C, T)) * 20 + 30 * np.sin(2 * np.pi * 50 * t) + 40 * np.sin(2 * np.pi * 130 * t) + 200
    taps = fir_design.preprocessing_filter_bank(s.preprocessing_filter, sfreq)
    notch = fir_design.notch_bank(sfreq, 50)
    eng = HotPathEngine(s, ["a", "b"], sfreq, notch_taps=notch, pre_taps=taps, lib=lib)
    starts = np.arange(6) * 100
    got = eng.process_batch(x, starts)
    opf, onf = orc.PreprocessingFilter(s, sfreq, taps=taps), orc.NotchFilter(sfreq, 50, taps=notch)
    for i, a in enumerate(starts):
        y = onf.process(opf.process(x[:, a:a + W]))
        want = {}
        for f in (orc.Hjorth(s, ["a", "b"], sfreq), orc.LineLength(s, ["a", "b"], sfreq)):
            want.update(f.calc_feature(y))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], [want[k] for k in eng.keys], s, sfreq,
                                       float(np.abs(x).max()), W)
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()


def case_config5_30khz_512pt(lib):
    """BASELINE config[4] shape at reduced channel count: 30 kHz, 512-sample windows, hop 30 samples
    (1 kHz feature rate), bands up to 7 kHz -- driven through the sample-based `window=` argument (the
    reference cannot express 512 samples: windowlength_ms is an integer).  FFT / STFT / Hjorth /
    LineLength / Raw / band-pass power (29 999-tap designs truncated to their live centre) / sharp
    waves vs the oracle.  Welch and bursts are excluded: at this rate the reference itself degenerates
    (nperseg = sfreq > W; samples_overlap = int(sfreq * seg_s / feat_hz) = 0)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    base = NMSettings.get_default().to_dict()
    base["frequency_ranges_hz"] = {"gamma": [60, 200], "HFA": [200, 500], "MUA": [500, 3000], "spike": [3000, 7000]}
    s = NMSettings(**base)
    for f in s.features.get_enabled():
        setattr(s.features, f, False)
    for f in ("fft", "stft", "raw_hjorth", "linelength", "return_raw", "bandpass_filter", "sharpwave_analysis"):
        setattr(s.features, f, True)
    s.sampling_rate_features_hz = 1000
    s.segment_length_features_ms = 17
    s.fft_settings.windowlength_ms = 17
    s.stft_settings.windowlength_ms = 17
    s.bandpass_filter_settings.segment_lengths_ms = {"gamma": 17, "HFA": 10, "MUA": 5, "spike": 3}
    # (a 60-500 Hz sharp-wave range is 60x oversampled here: neighbouring samples of the filtered
    # series differ by less than the fp32 error of the convolution and extrema move by a sample)
    s.sharpwave_analysis_settings.filter_ranges_hz = [[500, 3000], [1000, 7000]]
    s = NMSettings(**s.to_dict())
    sfreq, C, W, hop, nh = 30000.0, 6, 512, 30, 24
    rng = np.random.default_rng(0)
    T = W + (nh - 1) * hop
    t = np.arange(T) / sfreq
    x = rng.standard_normal((C, T)) * 30 + 40 * np.sin(2 * np.pi * 900 * t) + rng.uniform(-100, 100, (C, 1))
    ch = [f"c{i}" for i in range(C)]
    eng = HotPathEngine(s, ch, sfreq, lib=lib, window=W)
    got = eng.process_batch(x, np.arange(nh) * hop)
    feats = [orc._FEATURE_CLS[f](s, ch, sfreq) for f in eng.enabled]
    for i in range(nh):
        w = x[:, i * hop:i * hop + W]
        want = {}
        for f in feats:
            want.update(f.calc_feature(w))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], [want[k] for k in eng.keys], s, sfreq, 200.0, W,
                                       verifier=parity.Verifier(s, ch, sfreq, w))
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()


def case_config5_degenerate(lib):
    """BASELINE config[4]'s rate with the two features the plain C5 case leaves out, against the REFERENCE golden
    c5_degenerate.npz: Bursts with samples_overlap = 0 (the reference appends the whole window per hop, 40 hops
    across the overflow of a 15 000-sample ring), Welch under a shrunk segment (IndexError with bands beyond the
    shrunk spectrum; one low band: the reference's values at the frequencies the shrunk grid really has)."""
    import warnings

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import golden_dict, load_golden, settings_from_json

    g = load_golden("c5_degenerate")
    s = settings_from_json(g["settings_json"])
    sfreq, W, hop = float(g["sfreq"]), int(g["W"]), int(g["hop"])
    ch, data = [str(c) for c in g["ch_names"]], g["data"]
    s.features.disable_all()
    s.features.bursts = True
    keys = [str(k) for k in g["bursts_keys"]]
    want = g["bursts_values"]
    nh = len(want)
    eng = HotPathEngine(s, ch, sfreq, lib=lib, window=W, bank_taps=None)
    assert eng.keys == keys
    got = np.concatenate([eng.process_batch(data, np.arange(0, 13) * hop), eng.process_batch(data, np.arange(13, nh) * hop)])
    eng.close()
    for i in range(nh):
        w = data[:, i * hop:i * hop + W]
        n_bad, rep, _ = parity.compare(keys, got[i], want[i], s, sfreq, 60.0, W, verifier=parity.Verifier(s, ch, sfreq, w))
        assert n_bad == 0, f"bursts hop {i}\n{rep}"
    s.features.disable_all()
    s.features.welch = True
    with pytest.raises(IndexError):
        HotPathEngine(s, ch, sfreq, lib=lib, window=W)
    s1 = settings_from_json(g["welch1_settings_json"])
    s1.features.disable_all()
    s1.features.welch = True
    eng = HotPathEngine(s1, ch, sfreq, lib=lib, window=W)
    w1 = golden_dict(g, "welch1")
    assert eng.keys == list(w1)
    got1 = eng.process_window(data[:, :W])
    n_bad, rep, _ = parity.compare(eng.keys, got1, list(w1.values()), s1, sfreq, 60.0, W,
                                   verifier=parity.Verifier(s1, ch, sfreq, data[:, :W]))
    assert n_bad == 0, rep
    eng.close()


def case_raw_normalizer_order_methods(lib):
    """raw_normalization "median", "zscore-median" and the scikit-learn based "robust" / "minmax" vs the
    REFERENCE golden norm_methods.npz (14 windows, 0.7 s history: trims of 401 then 100 samples, a coarsely quantised
    channel with long runs of equal values): window by window through the drop-in class, and as two batches with the
    state carried through export / import (the sorted copy is rebuilt from the imported ring).  fp32 samples
    against float64 order statistics: 1e-5 relative + 2e-6 absolute."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine
    from py_neuromodulation_amd.processing import RawNormalizer
    from tests.helpers import load_golden

    g = load_golden("norm_methods")
    data, sfreq = g["raw_data"], 1000.0
    C = data.shape[0]
    for method in ("median", "zscore-median", "robust", "minmax", "quantile", "power"):
        s = NMSettings.get_default()
        s.raw_normalization_settings.normalization_time_s = 0.7
        s.raw_normalization_settings.normalization_method = method
        s.raw_normalization_settings.clip = 3
        starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
        want = g[f"raw_{method}"]
        rn = RawNormalizer(sfreq, s)
        mk = lambda feats=("return_raw",): HotPathEngine(NMSettings.get_default(), [f"c{i}" for i in range(C)], sfreq,   # noqa: E731
                                                          features=list(feats), raw_norm=rn._spec, window=1000, lib=lib)
        rn._engine = mk()
        for i, (a, b) in enumerate(zip(starts, ends)):
            y = rn.process(data[:, a:b])
            np.testing.assert_allclose(y[:, ::4], want[i], rtol=1e-5, atol=2e-6, err_msg=f"{method} hop {i}")
        rn._engine.close()
        # batches + state: the normalised LAST sample of each window is the return_raw feature
        e1 = mk()
        got = [e1.process_batch(data, starts[:6])]
        e2 = mk()
        e2.import_state(e1.export_state())
        got.append(e2.process_batch(data, starts[6:]))
        got = np.concatenate(got)
        ref = orc.RawNormalizer(sfreq, s)
        last = np.stack([ref.process(data[:, a:b])[:, -1] for a, b in zip(starts, ends)])
        np.testing.assert_allclose(got, last, rtol=1e-5, atol=2e-6, err_msg=f"{method} batched")
        e1.close(); e2.close()


def case_raw_quantile_subsample(lib):
    """raw_normalization "quantile" on a history of MORE than 10 000 samples: scikit-learn fits the 300 quantiles on a
    random 10 000-row subsample (random_state=None), so the reference's output is one draw of a random variable and
    cannot be matched value by value.  The device draws its own uniformly random 10 000-subset per hop and channel
    (hash keys + radix select, nmx_k_rawnorm.h).  Checked IN DISTRIBUTION against the float64 oracle run with 16
    different NumPy generators: the deviation from the exact (no subsampling) transform has the same size as the
    reference's own -- RMS within a factor 2 of the realisations' mean RMS, largest deviation within the
    realisations' largest x 2 -- and hops whose history is still <= 10 000 samples agree to 1e-5."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    sfreq, W, hop, C = 4000.0, 4000, 400, 2
    n_hops = 24
    rng = np.random.default_rng(5)
    data = (rng.standard_normal((C, W + (n_hops - 1) * hop)) * 40 + rng.uniform(-100, 100, (C, 1))).astype(np.float32)
    data[1] = np.round(data[1] / 2) * 2   # ties
    s = NMSettings.get_default()
    s.raw_normalization_settings.normalization_time_s = 3.2   # N = 12 800 samples
    s.raw_normalization_settings.normalization_method = "quantile"
    s.raw_normalization_settings.clip = 0
    starts = np.arange(n_hops) * hop
    spec = ("quantile", 0, int(3.2 * sfreq), int(sfreq / s.sampling_rate_features_hz))
    assert spec[3] == hop
    eng = HotPathEngine(NMSettings.get_default(), [f"c{i}" for i in range(C)], sfreq, features=["return_raw"],
                        raw_norm=spec, window=W, lib=lib)
    got = np.stack([eng.preprocess_window(data[:, a:a + W].astype(np.float64)) for a in starts])   # [hops][C][W]
    eng.close()

    def run(r):
        o = orc.RawNormalizer(sfreq, s, rng=r)
        return np.stack([o.process(data[:, a:a + W].astype(np.float64)) for a in starts])

    class _All:   # "subsample" = every row: the exact transform
        def choice(self, n, m, replace=False):
            return np.arange(n)
    exact = run(_All())
    hist_len = np.minimum(W + np.arange(n_hops) * hop, int(3.2 * sfreq) - 1 + hop)
    small = hist_len <= 10000
    assert small.any() and (~small).sum() >= 6
    np.testing.assert_allclose(got[small], exact[small], rtol=1e-5, atol=2e-6)
    refs = np.stack([run(np.random.default_rng(100 + k)) for k in range(16)])[:, ~small]
    dev = got[~small] - exact[~small]
    ref_dev = refs - exact[~small][None]
    rms_ref = np.sqrt((ref_dev ** 2).mean(axis=(1, 2, 3)))
    rms_dev = float(np.sqrt((dev ** 2).mean()))
    assert rms_ref.mean() > 0
    assert 0.5 * rms_ref.mean() <= rms_dev <= 2.0 * rms_ref.mean(), (rms_dev, rms_ref)
    assert np.abs(dev).max() <= 2.0 * np.abs(ref_dev).max() + 2e-6, (np.abs(dev).max(), np.abs(ref_dev).max())
    assert abs(dev.mean()) <= 4 * rms_dev / np.sqrt(dev.size / 50)   # no bias (values of a window are correlated)


def case_raw_normalizer(lib):
    """raw_normalization vs the reference golden (13 consecutive windows, 0.5 s history so the N - 1
    trim acts, zscore / mean / no clip): window-by-window through the drop-in class, as one batch with
    state carried through export/import, and in front of features.  Values are z-scores of fp32 data
    against float64 statistics: 1e-5 relative + 2e-6 absolute."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine
    from py_neuromodulation_amd.processing import RawNormalizer
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("raw_normalizer")
    sfreq, data, st = float(g["sfreq"]), g["data"], int(g["stride"])
    C = data.shape[0]
    for tag in ("zscore", "mean", "zscore_noclip"):
        s = settings_from_json(g[f"{tag}_settings_json"])
        starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz,
                                              s.segment_length_features_ms)
        want = g[f"{tag}_y"]
        rn = RawNormalizer(sfreq, s)
        rn._engine = HotPathEngine(NMSettings.get_default(), [f"c{i}" for i in range(C)], sfreq,
                                   features=["return_raw"], raw_norm=rn._spec, window=1000, lib=lib)
        for i, (a, b) in enumerate(zip(starts, ends)):
            y = rn.process(data[:, a:b])
            np.testing.assert_allclose(y[:, ::st], want[i], rtol=1e-5, atol=2e-6, err_msg=f"{tag} hop {i}")
    # batch + state: the normalised LAST sample of each window is the return_raw feature
    s = settings_from_json(g["zscore_settings_json"])
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    spec = ("zscore", 3, int(0.5 * sfreq), int(sfreq / s.sampling_rate_features_hz))
    mk = lambda: HotPathEngine(NMSettings.get_default(), [f"c{i}" for i in range(C)], sfreq,   # noqa: E731
                               features=["return_raw", "raw_hjorth"], raw_norm=spec, window=1000, lib=lib)
    ref = orc.RawNormalizer(sfreq, s)
    hj, rw = orc.Hjorth(s, [f"c{i}" for i in range(C)], sfreq), orc.Raw(s, [f"c{i}" for i in range(C)], sfreq)
    rows = []
    for a, b in zip(starts, ends):
        y = ref.process(data[:, a:b])
        d = hj.calc_feature(y)
        d.update(rw.calc_feature(y))
        rows.append(d)
    e1 = mk()
    got = [e1.process_batch(data, starts[:5])]
    e2 = mk()
    e2.import_state(e1.export_state())
    got.append(e2.process_batch(data, starts[5:]))
    got = np.concatenate(got)
    for i in range(len(starts)):
        np.testing.assert_allclose(got[i], [rows[i][k] for k in e1.keys], rtol=2e-5, atol=5e-6, err_msg=f"hop {i}")
    e2.reset_state()
    np.testing.assert_allclose(e2.process_batch(data, starts[:2]), got[:2], rtol=1e-6, atol=1e-7)
    e1.close()
    e2.close()
    # an artefact 1e6 x the signal passes through the 0.5 s history: the sliding sums are rebuilt once it has left
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 1000 + 20 * 100)) * 1e-3 + 0.01
    x[0, 1100:1110] = 5e3
    st2 = np.arange(21) * 100
    for method in ("zscore", "zscore-median"):
        s2 = settings_from_json(g["zscore_settings_json"])
        s2.raw_normalization_settings.normalization_method = method
        s2.raw_normalization_settings.clip = 0
        ref = orc.RawNormalizer(sfreq, s2)
        want = np.stack([ref.process(x[:, a:a + 1000])[:, -1] for a in st2])
        eng = HotPathEngine(NMSettings.get_default(), ["a", "b"], sfreq, features=["return_raw"],
                            raw_norm=(method, 0, int(0.5 * sfreq), 100), window=1000, lib=lib)
        got = eng.process_batch(x, st2)
        np.testing.assert_allclose(got[1:], want[1:], rtol=2e-5, atol=5e-6, err_msg=f"artefact, {method}")
        eng.close()


def case_psd_keys_skip_normalisation(lib):
    """return_spectrum adds "<ch>_fft_psd_<f>" keys; with normalize_psd = False (default) the reference
    normalises every other column and passes the psd ones through (stream/data_processor.py:263-290).
    Stream.run (device normaliser with a column mask) vs the oracle's hop-by-hop DataProcessor."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    for f in s.features.get_enabled():
        setattr(s.features, f, False)
    s.features.fft = True
    s.features.raw_hjorth = True
    s.fft_settings.return_spectrum = True
    s.preprocessing = ["re_referencing"]
    s.feature_normalization_settings.normalization_time_s = 1.0   # 10 rows: the N - 1 trim acts
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 4000)) * 10 + 100
    st = Stream(sfreq=1000.0, data=x, settings=s, lib=lib)
    df = st.run(x, save_csv=False)
    chd = st.channels.to_dict("list")
    # stage 1: un-normalised features (psd keys included) vs the oracle, standard policy
    s_raw = type(s)(**s.to_dict())
    s_raw.postprocessing.feature_normalization = False
    raw = Stream(sfreq=1000.0, data=x, settings=s_raw, lib=lib).run(x, save_csv=False)
    rows = orc.run_stream(x, 1000.0, s_raw, channels=chd)
    assert list(df.columns) == list(raw.columns) == list(rows[0].keys())
    keys = list(df.columns)[:-1]
    starts, ends, _ = orc.window_schedule(x.shape[1], 1000.0, s.sampling_rate_features_hz, s.segment_length_features_ms)
    pv = parity.PipelineVerifiers(s_raw, chd, 1000.0, x, starts, 1000)
    rawv = raw.to_numpy(dtype=np.float64)[:, :-1]
    for r in range(len(rows)):
        n_bad, rep, _ = parity.compare(keys, rawv[r], [rows[r][k] for k in keys], s_raw, 1000.0, 140.0, 1000,
                                       verifier=pv.row(r))
        assert n_bad == 0, f"row {r}\n{rep}"
    # stage 2: the device normaliser (with its psd column mask) vs the oracle normaliser on the same rows
    psd = np.array(["psd" in c for c in keys])
    assert psd.sum() == 3 * 501
    norm = orc.FeatureNormalizer(s)
    want = rawv.copy()
    for r in range(len(want)):
        want[r, ~psd] = norm.process(rawv[r, ~psd].copy())
    got = df.to_numpy(dtype=np.float64)[:, :-1]
    np.testing.assert_array_equal(got[:, psd], rawv[:, psd])
    np.testing.assert_allclose(got[:, ~psd], want[:, ~psd], rtol=1e-5, atol=2e-6)


def case_reref_structured_matrices(lib, monkeypatch=None):
    """Re-reference matrices other than the full common average -- row subsets of one (a channel shard of
    a jointly referenced array), two type groups, bipolar rows, the mixed table of the reference golden,
    and an unstructured (random) matrix that must fall back to the dense product -- against R @ nan_to_num(x)
    in float64 (processing/rereference.py:88-102).  fp32 output of float64 sums: 1e-6 * amplitude."""
    import json

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import load_golden

    s = NMSettings.get_default()
    rng = np.random.default_rng(3)
    W = 1000
    cases = []
    n = 40
    car = np.full((n, n), -1.0 / (n - 1))
    np.fill_diagonal(car, 1.0)
    cases.append(("shard of a common average", car[8:24]))
    two = np.zeros((n, n))
    for lo, hi in ((0, 25), (25, 40)):          # two type groups, each its own average
        m = hi - lo
        two[lo:hi, lo:hi] = -1.0 / (m - 1)
    np.fill_diagonal(two, 1.0)
    two[3] = 0; two[3, 3] = 1; two[3, 30] = -1               # a bipolar row
    two[4] = 0; two[4, 4] = 1; two[4, [5, 6]] = -0.5         # a row referenced to two channels
    cases.append(("two groups + bipolar rows", two))
    cases.append(("rows of both groups, bad column", np.delete(two[[0, 3, 26, 39, 4]], 7, axis=1)))
    g = load_golden("pipeline_nan_channels")
    cases.append(("reference golden: mixed table", np.asarray(g["mix_ref_matrix"], np.float64)))
    cases.append(("unstructured", rng.standard_normal((6, 9))))
    for name, R in cases:
        C, Cin = R.shape
        x = rng.standard_normal((Cin, W)) * 40 + rng.uniform(-400, 400, (Cin, 1))
        x[1, 17] = np.nan
        eng = HotPathEngine(s, [f"c{i}" for i in range(C)], 1000.0, lib=lib, features=["return_raw"], ref_matrix=R)
        got = eng.preprocess_window(x)
        want = R @ np.nan_to_num(x)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6 * np.abs(x[~np.isnan(x)]).max(), err_msg=name)
        eng.close()


def case_resampler_long_windows(lib):
    """raw_resampling of windows whose padded length (8192 < n_pad <= 65 536) does not fit one LDS transform -- 8 / 10 /
    24 / 30 / 44.1 kHz recordings with 1 s windows brought to 1 kHz (the reference's DEFAULT pre-processing), one with
    an ODD resampled padded length (24 kHz: round(32768 / 24) = 1365, full complex inverse) -- through the polyphase
    path of nmx_k_resample.h, against the restated mne.filter.resample.  NMX_RESAMPLE_POLY=1 routes a short window
    (4 kHz) through the same path: both paths must agree with the oracle.  Then Stream.run at 10 kHz with
    resample_features_at_new_rate=True against the oracle's pipeline.  Tolerance as case_resampler (2e-5 x amplitude)."""
    import os

    from oracle import mne_restated as mr
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.engine import HotPathEngine
    from py_neuromodulation_amd.stream import Stream

    rng = np.random.default_rng(23)
    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.return_raw = True
    s.postprocessing.feature_normalization = False
    for sf_old, sf_new, W, poly in ((8000, 1000, 8000, 0), (10000, 1000, 10000, 0), (24000, 1000, 24000, 0),
                                    (30000, 1000, 30000, 0), (44100, 1000, 44100, 0), (30000, 2000, 15000, 0),
                                    (4000, 1000, 4000, 1)):
        t = np.arange(W) / sf_old
        x = rng.standard_normal((2, W)) * 20 + 50 * np.sin(2 * np.pi * 11 * t) + rng.uniform(-300, 300, (2, 1))
        x[1, W // 3] = np.nan
        if poly:
            os.environ["NMX_RESAMPLE_POLY"] = "1"
        try:
            eng = HotPathEngine(s, ["a", "b"], float(sf_new), resample_from=float(sf_old), raw_window=W,
                                window=int(round(sf_new / sf_old * W)), lib=lib)
        finally:
            os.environ.pop("NMX_RESAMPLE_POLY", None)
        want = mr.resample(np.nan_to_num(x), up=sf_new / sf_old, down=1.0)
        got = eng.preprocess_window(x)
        assert got.shape == want.shape == (2, int(round(sf_new / sf_old * W)))
        amp = np.abs(np.nan_to_num(x)).max()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * amp, err_msg=f"{sf_old}->{sf_new}")
        eng.close()
    # Stream.run on a 10 kHz recording: notch at the raw rate (9999 taps: partitioned overlap-save), resampling to 1 kHz,
    # common average, features designed for 1 kHz
    from py_neuromodulation_amd import fir_design

    s.features.fft = True
    s.features.raw_hjorth = True
    s.preprocessing = ["raw_resampling", "notch_filter", "re_referencing"]
    sf, W = 10000.0, 10000
    T = W + 3 * 1000 + 3
    t = np.arange(T) / sf
    xs = rng.standard_normal((2, T)) * 10 + 20 * np.sin(2 * np.pi * 21 * t) + 6 * np.sin(2 * np.pi * 50 * t)
    st = Stream(sfreq=sf, data=xs, settings=s, line_noise=50, lib=lib, resample_features_at_new_rate=True)
    df = st.run(xs, save_csv=False)
    assert len(df) == 4 and st.data_processor.sfreq_raw == 1000.0
    nf = orc.NotchFilter(sf, 50, taps=fir_design.notch_bank(sf, 50))
    R = np.array([[1.0, -1.0], [-1.0, 1.0]])
    names = [k for k in df.columns if k != "time"]
    chn = ["ch0_avgref", "ch1_avgref"]
    for i in range(4):
        w = xs[:, i * 1000:i * 1000 + W]
        y = R @ mr.resample(nf.process(w), up=0.1, down=1.0)
        want = {}
        for f in (orc.Hjorth(s, chn, 1000.0), orc.Raw(s, chn, 1000.0), orc.FFT(s, chn, 1000.0)):
            want.update(f.calc_feature(y))
        n_bad, rep, _ = parity.compare(names, df.iloc[i][names].to_numpy(dtype=np.float64), [want[k] for k in names], s,
                                       1000.0, float(np.abs(xs).max()), 1000,
                                       verifier=parity.Verifier(s, chn, 1000.0, y, raw=w))
        assert n_bad == 0, f"stream hop {i}\n{rep}"


def case_raw_resampling_reference_quirk(lib):
    """Default settings on recordings that are not sampled at resample_freq_hz: the reference resamples each
    window but keeps designing notch AND features with the raw rate (stream/data_processor.py:55,68,80;
    processing/resample.py:36-60).  Stream.run must reproduce that -- golden from the reference's own
    Stream.run at 2 kHz (down-sampling: FFT / Welch segments longer than the window, clamped band-pass
    tails, seg_s of the bursts) and 250 Hz (up-sampling: seven Welch segments, FFT of the window's tail)."""
    import json

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("resample_quirk")
    for tag in ("down_2k", "up_250"):
        sf = float(g[f"{tag}_sfreq"])
        data = g[f"{tag}_data"]
        s = settings_from_json(g[f"{tag}_settings_json"])
        st = Stream(sfreq=sf, data=data, settings=s, line_noise=50, lib=lib)
        df = st.run(save_csv=False)
        assert st.data_processor.sfreq_raw == sf          # like the reference: the raw rate stays
        cols = [str(c) for c in g[f"{tag}_columns"]]
        assert list(df.columns) == cols
        got, want = df.to_numpy(dtype=np.float64), g[f"{tag}_values"]
        assert got.shape == want.shape
        np.testing.assert_array_equal(got[:, -1], want[:, -1])
        ch = json.loads(str(g[f"{tag}_channels_json"]))
        starts, ends, _ = orc.window_schedule(data.shape[1], sf, s.sampling_rate_features_hz, s.segment_length_features_ms)
        pv = parity.PipelineVerifiers(s, ch, sf, data, starts, int(ends[0] - starts[0]), line_noise=50, ends=ends)
        W_new = int(round(1000.0 / sf * (ends[0] - starts[0])))
        for r in range(len(got)):
            n_bad, rep, _ = parity.compare(cols[:-1], got[r, :-1], want[r, :-1], s, sf, float(np.abs(data).max()), W_new,
                                           verifier=pv.row(r))
            assert n_bad == 0, f"{tag} row {r}\n{rep}"


RANDOM_SETTINGS_SEEDS = list(range(101, 141))


def random_settings(seed):
    """A seeded random point of the settings space of the hot path: sampling rate, window, feature rate, channel
    count, band set, feature families, pre-processing.  Returns (settings, sfreq, data, line_noise)."""
    from py_neuromodulation_amd import NMSettings

    rng = np.random.default_rng(seed)
    for _ in range(50):
        sfreq = float(rng.choice([250, 400, 500, 512, 750, 1000, 1200, 2000, 4000]))
        seg_ms = int(rng.choice([250, 500, 1000, 1500]))
        feat_hz = int(rng.choice([5, 10, 20]))
        n_ch = int(rng.integers(1, 8))
        nyq = sfreq / 2
        pool = [("theta", [4, 8]), ("alpha", [8, 12]), ("low_beta", [13, 20]), ("high_beta", [20, 35]),
                ("low_gamma", [60, 80]), ("high_gamma", [90, 200]), ("HFA", [200, 400])]
        fit = [(n, r) for n, r in pool if r[1] + 10 < nyq]
        keep = [b for b in fit if rng.random() < 0.7]
        if len(keep) < 2:
            keep = fit[:2]
        base = NMSettings.get_default().to_dict()
        base["frequency_ranges_hz"] = {n: r for n, r in keep}
        s = NMSettings(**base)
        s.sampling_rate_features_hz = feat_hz
        s.segment_length_features_ms = seg_ms
        s.features.disable_all()
        fams = ["fft", "welch", "stft", "bandpass_filter", "raw_hjorth", "linelength", "return_raw",
                "sharpwave_analysis", "bursts"]
        on = [f for f in fams if rng.random() < 0.55]
        if not on:
            on = ["fft"]
        if seg_ms < 500 or sfreq * seg_ms / 1000 > 14000:   # (the sharp-wave kernel takes windows up to ~14 500 samples: LDS)
            on = [f for f in on if f != "sharpwave_analysis"] or ["fft"]
        # (ragged window LENGTHS -- a non-integer segment -- run one plan per length; the burst history and the Kalman
        # filters travel between them: case_ragged_bursts)
        for f in on:
            setattr(s.features, f, True)
        for name in ("fft_settings", "welch_settings", "stft_settings"):
            s[name].windowlength_ms = int(min(s[name].windowlength_ms, seg_ms))
        seg = {}
        for n, _ in keep:
            seg[n] = int(min(s.bandpass_filter_settings.segment_lengths_ms.get(n, seg_ms), seg_ms))
        s.bandpass_filter_settings.segment_lengths_ms = seg
        s.bandpass_filter_settings.log_transform = bool(rng.random() < 0.7)
        s.bandpass_filter_settings.bandpower_features.activity = True
        s.bandpass_filter_settings.bandpower_features.mobility = bool(rng.random() < 0.3)
        s.bandpass_filter_settings.bandpower_features.complexity = bool(rng.random() < 0.3)
        burst_bands = [n for n, _ in keep if rng.random() < 0.5] or [keep[0][0]]
        s.bursts_settings.frequency_bands = burst_bands
        s.bursts_settings.time_duration_s = float(rng.choice([3, 10, 30]))
        s.bursts_settings.threshold = float(rng.choice([60, 75, 90]))
        pre = []
        if rng.random() < 0.4:
            pre.append("notch_filter")
        if rng.random() < 0.5 and n_ch >= 2:
            pre.append("re_referencing")
        s.preprocessing = pre
        s.postprocessing.feature_normalization = False
        try:
            s = NMSettings(**s.to_dict()).validate()
        except Exception:
            continue
        W = int(sfreq * seg_ms / 1000)
        n_hops = int(rng.integers(6, 14))
        T = W + int(n_hops * sfreq / feat_hz) + int(rng.integers(0, 7))
        t = np.arange(T) / sfreq
        data = rng.standard_normal((n_ch, T)) * 10 + rng.uniform(-50, 50, (n_ch, 1))
        data += 8 * np.sin(2 * np.pi * rng.uniform(5, min(40, nyq / 3)) * t)[None]
        line = 50 if sfreq > 130 else None
        return s, sfreq, data, line
    raise RuntimeError("no valid settings drawn")


def random_settings_highrate(seed):
    """Random settings at 6 / 8 kHz with 1 s windows and NO resampling -- the shapes of round 3's partitioned overlap-save
    (or at 8 - 24 kHz WITH the default raw_resampling to 1 kHz: polyphase resampler behind a partitioned notch)
    FIR mode (band-pass taps of ~10 000 - 13 000, notch taps of 5 999 / 7 999) -- with or without notch, re-referencing,
    bursts (stand-alone Hilbert kernel behind the partitioned bank), a raw "zscore" normaliser.  Returns (settings, sfreq, data, line_noise)."""
    from py_neuromodulation_amd import NMSettings

    rng = np.random.default_rng(70_000 + seed)
    resample = rng.random() < 0.4   # the reference's default: windows brought to 1 kHz, designs at the RAW rate (quirk)
    sfreq = float(rng.choice([8000, 10000, 16000, 24000]) if resample else rng.choice([6000, 8000]))
    n_ch = int(rng.choice([1, 2, 3]))
    pool = [("theta", [4, 8]), ("alpha", [8, 12]), ("low_beta", [13, 20]), ("high_beta", [20, 35]),
            ("low_gamma", [60, 80]), ("high_gamma", [90, 200])]
    keep = [b for b in pool if rng.random() < 0.5]
    if len(keep) < 2:
        keep = pool[2:4]
    base = NMSettings.get_default().to_dict()
    base["frequency_ranges_hz"] = {n: r for n, r in keep}
    s = NMSettings(**base)
    s.features.disable_all()
    on = [f for f in ("fft", "bandpass_filter", "raw_hjorth", "linelength", "return_raw", "bursts") if rng.random() < 0.6]
    if "bandpass_filter" not in on and "bursts" not in on:
        on.append("bandpass_filter")
    for f in on:
        setattr(s.features, f, True)
    s.bandpass_filter_settings.segment_lengths_ms = {n: int(rng.choice([100, 333, 500, 1000])) for n, _ in keep}
    s.bursts_settings.frequency_bands = [n for n, _ in keep][:int(rng.integers(1, 3))]
    s.bursts_settings.time_duration_s = float(rng.choice([1, 2]))
    s.bursts_settings.threshold = float(rng.choice([50, 75, 90]))
    pre = ["raw_resampling"] if resample else []
    if rng.random() < 0.6:
        pre.append("notch_filter")
    if rng.random() < 0.5 and n_ch >= 2:
        pre.append("re_referencing")
    if rng.random() < 0.3 and not resample:
        pre.append("raw_normalization")
        # (the order-statistic raw normalisers keep window + hop <= 6484 samples in LDS: INTEGRATION.md section 4)
        # ("mean" divides by the history's MEAN: on a channel whose offset is near zero every feature is the fp32 error
        # of that mean blown up -- 1 of 10 500 sweep cases, 2e-4 on a Hjorth activity of 8.5e6 -- and nothing verifies it)
        s.raw_normalization_settings.normalization_method = str(rng.choice(["zscore", "zscore"]))
        s.raw_normalization_settings.normalization_time_s = float(rng.choice([0.3, 1.0]))
        # a single z-scored sample (`return_raw`) carries the fp32 error of the 8 000-tap notch divided by the history's
        # spread -- |z| = 24 and 4e-6 relative in two of 14 000 sweep cases -- and has no conditioning verifier: left out
        s.features.return_raw = False
        if not s.features.get_enabled():
            s.features.linelength = True
        s.raw_normalization_settings.clip = float(rng.choice([0, 3]))
    s.preprocessing = pre
    s.postprocessing.feature_normalization = False
    s = NMSettings(**s.to_dict()).validate()
    W, hop = int(sfreq), int(sfreq / 10)
    T = W + int(rng.integers(2, 5)) * hop + int(rng.integers(0, 5))
    t = np.arange(T) / sfreq
    data = rng.standard_normal((n_ch, T)) * 10 + rng.uniform(-50, 50, (n_ch, 1))
    data += 8 * np.sin(2 * np.pi * rng.uniform(5, 40) * t)[None] + 4 * np.sin(2 * np.pi * 50 * t)[None]
    return s, sfreq, data, 50


def case_random_settings_highrate(lib, seed):
    """`random_settings_highrate` through `Stream.run` against the oracle's `run_stream` (as case_random_settings)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.stream import Stream

    s, sfreq, data, line = random_settings_highrate(seed)
    ch = chmod.get_default_channels_from_data(data).to_dict("list")
    df = Stream(sfreq, data=data, settings=s, line_noise=line, lib=lib).run(save_csv=False)
    rows = orc.run_stream(data, sfreq, s, ch, line_noise=line)
    assert list(df.columns) == list(rows[0].keys()), f"seed {seed}: columns differ"
    got = df.to_numpy(float)
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    W = int(ends[0] - starts[0])
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, W, line_noise=line, ends=ends)
    cols = list(df.columns)
    for i, r in enumerate(rows):
        want = np.array(list(r.values()))
        n_bad, rep, _ = parity.compare(cols[:-1], got[i, :-1], want[:-1], s, sfreq, 40.0, W, verifier=pv.row(i))
        assert n_bad == 0, f"seed {seed} ({sfreq} Hz, {data.shape[0]} ch, {s.features.get_enabled()}, {s.preprocessing}) hop {i}\n{rep}"


def case_random_settings(lib, seed):
    """Drop-in check away from the benchmark shapes: a random valid settings object / sampling rate / channel count
    through `Stream.run`, against the oracle's `run_stream` on the same recording, column order included."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.stream import Stream

    s, sfreq, data, line = random_settings(seed)
    ch = chmod.get_default_channels_from_data(data).to_dict("list")
    try:
        df = Stream(sfreq, data=data, settings=s, line_noise=line, lib=lib).run(save_csv=False)
    except (ValueError, IndexError) as e:
        # a combination the reference cannot run either (e.g. Welch on a window shorter than one second:
        # features/oscillatory.py builds its frequency grid for nperseg = sfreq): the error is the parity
        with pytest.raises(Exception):
            orc.run_stream(data, sfreq, s, ch, line_noise=line)
        return f"both raise: {e}"
    rows = orc.run_stream(data, sfreq, s, ch, line_noise=line)
    assert list(df.columns) == list(rows[0].keys()), f"seed {seed}: columns differ"
    assert len(df) == len(rows)
    got = df.to_numpy(float)
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    W = int(ends[0] - starts[0])
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, W, line_noise=line, ends=ends)
    cols = list(df.columns)
    for i, r in enumerate(rows):
        want = np.array(list(r.values()))
        n_bad, rep, _ = parity.compare(cols[:-1], got[i, :-1], want[:-1], s, sfreq, 40.0, W, verifier=pv.row(i))
        assert n_bad == 0, f"seed {seed} ({sfreq} Hz, W {W}, {data.shape[0]} ch) hop {i}\n{rep}"
        assert got[i, -1] == want[-1]


def case_high_rate_partitioned_fir(lib):
    """A recording at 8 kHz with 1 s windows and NO resampling: the automatic band-pass taps are 13 201 long, the notch
    7 999 -- the FFT convolution of a window (M >= 14 600 / 16 000) does not fit one LDS transform, so notch and
    band-pass bank run as partitioned overlap-save convolutions (nmx_k_bank.h: NmxBankArgs::partitioned; round 2 refused this shape), the
    burst bands go through the stand-alone Hilbert kernel.  Stream.run against the oracle's run_stream."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.stream import Stream

    sfreq, C = 8000.0, 2
    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = s.features.raw_hjorth = s.features.bursts = True
    s.frequency_ranges_hz = {"alpha": [8, 12], "high_beta": [20, 35]}
    s.bandpass_filter_settings.segment_lengths_ms = {"alpha": 500, "high_beta": 333}
    s.bursts_settings.frequency_bands = ["high_beta"]
    s.bursts_settings.time_duration_s = 3
    s.preprocessing = ["notch_filter"]
    s.postprocessing.feature_normalization = False
    s = s.validate() if hasattr(s, "validate") else s
    rng = np.random.default_rng(8)
    T = 8000 + 2 * 800
    t = np.arange(T) / sfreq
    data = rng.standard_normal((C, T)) * 30 + 15 * np.sin(2 * np.pi * 25 * t) + 8 * np.sin(2 * np.pi * 50 * t) + rng.uniform(-80, 80, (C, 1))
    ch = chmod.get_default_channels_from_data(data).to_dict("list")
    df = Stream(sfreq, data=data, settings=s, line_noise=50, lib=lib).run(save_csv=False)
    rows = orc.run_stream(data, sfreq, s, ch, line_noise=50)
    assert list(df.columns) == list(rows[0].keys()) and len(df) == len(rows) == 3
    got = df.to_numpy(float)
    starts, ends, _ = orc.window_schedule(T, sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    W = int(ends[0] - starts[0])
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, W, line_noise=50, ends=ends)
    cols = list(df.columns)
    for i, r in enumerate(rows):
        want = np.array(list(r.values()))
        n_bad, rep, _ = parity.compare(cols[:-1], got[i, :-1], want[:-1], s, sfreq, 40.0, W, verifier=pv.row(i))
        assert n_bad == 0, f"hop {i}\n{rep}"


def case_short_windows(lib):
    """Windows shorter than the nominal Welch / STFT segment (tests/golden/short_windows.npz, produced by the
    reference): scipy shrinks nperseg, the band bins keep the nominal grid -> other frequencies, IndexError or
    ValueError; the engine reproduces all three."""
    from py_neuromodulation_amd.engine import HotPathEngine
    from tests.helpers import golden_dict, load_golden, settings_from_json

    g = load_golden("short_windows")
    ch = ["c0", "c1", "c2"]
    for tag in "abcd":
        x, sfreq = g[f"{tag}_data"], float(g[f"{tag}_sfreq"])
        for fam in ("fft", "welch", "stft"):
            s = settings_from_json(g[f"{tag}_settings_json"])
            s.features.disable_all()
            setattr(s.features, fam, True)
            err = str(g[f"{tag}_{fam}_error"])
            if err:
                with pytest.raises(Exception) as ei:
                    HotPathEngine(s, ch, sfreq, window=x.shape[1], lib=lib)
                assert type(ei.value).__name__ == err, f"{tag} {fam}: {ei.value!r}"
                continue
            eng = HotPathEngine(s, ch, sfreq, window=x.shape[1], lib=lib)
            got = eng.process_window(x)
            want = golden_dict(g, f"{tag}_{fam}")
            assert eng.keys == list(want), f"{tag} {fam}"
            n_bad, rep, _ = parity.compare(eng.keys, got, list(want.values()), s, sfreq, 50.0, x.shape[1],
                                           verifier=parity.Verifier(s, ch, sfreq, x))
            assert n_bad == 0, f"{tag} {fam}\n{rep}"
            eng.close()


WIDE_SETTINGS_SEEDS = [i for i in range(201, 242) if i != 216]   # (216: four 3999-tap stages x 34 channels, minutes in the oracle)


def random_settings_wide(seed):
    """Second generator: everything `random_settings` draws plus estimator sets, return_spectrum, sharp-wave feature /
    estimator tables, Kalman smoothing, raw_resampling / preprocessing_filter, channel counts up to 70, a flat channel,
    and a feature-normalisation method.  Returns (settings, sfreq, data, line_noise, norm_method or None)."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.settings import _SW_FEATURES

    rng = np.random.default_rng(seed)
    for _ in range(200):
        sfreq = float(rng.choice([250, 500, 600, 1000, 1000, 2000, 2048, 4000]))
        seg_ms = int(rng.choice([500, 1000, 1000, 2000]))
        feat_hz = int(rng.choice([4, 10, 10, 20]))
        n_ch = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 64, 70]))
        pre = []
        fs_feat = sfreq
        if rng.random() < 0.3 and sfreq * seg_ms / 1000 <= 7992:   # (the resample kernel pads to <= 8192 samples)
            rs = float(rng.choice([250, 500, 1000]))
            if rs < sfreq:
                pre.append("raw_resampling")
                fs_feat = rs
        nyq = fs_feat / 2
        pool = [("theta", [4, 8]), ("alpha", [8, 12]), ("low_beta", [13, 20]), ("high_beta", [20, 35]),
                ("low_gamma", [60, 80]), ("high_gamma", [90, 200]), ("HFA", [200, 400])]
        fit = [(n, r) for n, r in pool if r[1] + 10 < nyq]
        keep = [b for b in fit if rng.random() < 0.6]
        if len(keep) < 2:
            keep = fit[:2]
        base = NMSettings.get_default().to_dict()
        base["frequency_ranges_hz"] = {n: r for n, r in keep}
        s = NMSettings(**base)
        s.sampling_rate_features_hz = feat_hz
        s.segment_length_features_ms = seg_ms
        if "raw_resampling" in pre:
            s.raw_resampling_settings.resample_freq_hz = fs_feat
        s.features.disable_all()
        fams = ["fft", "welch", "stft", "bandpass_filter", "raw_hjorth", "linelength", "return_raw",
                "sharpwave_analysis", "bursts"]
        on = [f for f in fams if rng.random() < 0.5] or ["fft"]
        W = int(sfreq * seg_ms / 1000)
        if W > 14000:
            on = [f for f in on if f != "sharpwave_analysis"] or ["fft"]
        if seg_ms < 1000 and "welch" in on:
            on = [f for f in on if f != "welch"] or ["fft"]
        if (sfreq * seg_ms / 1000) % 1 or "raw_resampling" in pre or W > 6000:   # (W > 6000: the Hilbert stage of the generic
            on = [f for f in on if f != "bursts"] or ["fft"]                 # bank kernel needs 2 x 8 W bytes of LDS)
        for f in on:
            setattr(s.features, f, True)
        for name in ("fft_settings", "welch_settings", "stft_settings"):
            o = s[name]
            o.windowlength_ms = int(min(rng.choice([250, 500, 1000]), seg_ms))
            ests = [e for e in ("mean", "median", "std", "max") if rng.random() < 0.4] or ["mean"]
            for e in ("mean", "median", "std", "max"):
                setattr(o.features, e, e in ests)
            o.log_transform = bool(rng.random() < 0.8)
            # (a bin spacing below 1 Hz makes the reference's 'psd_<int(f)>' keys collide: not supported, raises)
            spacing = sfreq / min(o.windowlength_ms, W) if name == "stft_settings" else 1000.0 / o.windowlength_ms
            o.return_spectrum = bool(rng.random() < 0.15) and spacing >= 1.0 and name != "welch_settings"
        s.bandpass_filter_settings.segment_lengths_ms = {
            n: int(min(rng.choice([100, 333, 500, 1000]), seg_ms)) for n, _ in keep}
        s.bandpass_filter_settings.log_transform = bool(rng.random() < 0.7)
        for f in ("activity", "mobility", "complexity"):
            setattr(s.bandpass_filter_settings.bandpower_features, f, bool(rng.random() < 0.5))
        if not s.bandpass_filter_settings.bandpower_features.get_enabled():
            s.bandpass_filter_settings.bandpower_features.activity = True
        if rng.random() < 0.25 and s.bandpass_filter_settings.bandpower_features.activity:
            s.bandpass_filter_settings.kalman_filter = True
            s.kalman_filter_settings.frequency_bands = [n for n, _ in keep if rng.random() < 0.6] or [keep[0][0]]
        s.bursts_settings.frequency_bands = [n for n, _ in keep if rng.random() < 0.5] or [keep[0][0]]
        s.bursts_settings.time_duration_s = float(rng.choice([1, 3, 30]))
        s.bursts_settings.threshold = float(rng.choice([50, 75, 90]))
        for f in ("duration", "amplitude", "burst_rate_per_s", "in_burst"):
            setattr(s.bursts_settings.burst_features, f, bool(rng.random() < 0.7))
        if not s.bursts_settings.burst_features.get_enabled():
            s.bursts_settings.burst_features.duration = True
        sw = s.sharpwave_analysis_settings
        feats = [f for f in _SW_FEATURES if rng.random() < 0.3] or ["prominence"]
        for f in _SW_FEATURES:
            setattr(sw.sharpwave_features, f, f in feats)
        est = {e: [] for e in ("mean", "median", "max", "min", "var")}
        for f in feats:
            for e in rng.choice(list(est), size=int(rng.integers(1, 3)), replace=False):
                est[str(e)].append(f)
        for e, fl in est.items():
            sw.estimator[e] = fl
        sw.apply_estimator_between_peaks_and_troughs = bool(rng.random() < 0.5)
        sw.detect_troughs.estimate = True
        sw.detect_peaks.estimate = bool(sw.apply_estimator_between_peaks_and_troughs or rng.random() < 0.5)
        for node in (sw.detect_troughs, sw.detect_peaks):
            node.distance_troughs_ms = float(rng.choice([3, 5, 10, 15]))
            node.distance_peaks_ms = float(rng.choice([3, 5, 10, 15]))
        hi = 80 if nyq > 100 else 40
        sw.filter_ranges_hz = [[[5, hi], [5, 30]], [[5, 30]], [[10, hi]]][int(rng.integers(0, 3))]
        if rng.random() < 0.4:
            pre.append("notch_filter")
        if rng.random() < 0.5 and n_ch >= 2:
            pre.append("re_referencing")
        if rng.random() < 0.2 and fs_feat >= 1000 and "raw_resampling" not in pre:
            pre.append("preprocessing_filter")
            for f in ("bandstop_filter", "bandpass_filter", "lowpass_filter", "highpass_filter"):
                setattr(s.preprocessing_filter, f, bool(rng.random() < 0.5))
        # (window + hop beyond 6484 samples: the order-statistic normalisers keep their merge lists in device memory)
        raw_norm = bool(rng.random() < 0.15) and "raw_resampling" not in pre
        if raw_norm:
            pre.append("raw_normalization")
            # ("mean" / "median" divide by the centre: ill-conditioned on re-referenced, near-zero-mean rows; they are
            # pinned by the reference goldens of case_raw_normalizer* on data with an offset)
            s.raw_normalization_settings.normalization_method = str(
                rng.choice(["zscore", "zscore-median", "robust", "minmax"]))
            s.raw_normalization_settings.normalization_time_s = float(rng.choice([1, 3, 30]))
            s.raw_normalization_settings.clip = float(rng.choice([0, 3]))
        s.preprocessing = pre
        s.postprocessing.feature_normalization = False
        norm = None
        if rng.random() < 0.5:
            norm = str(rng.choice(["zscore", "mean", "median", "zscore-median", "robust", "minmax", "quantile"]))
            s.feature_normalization_settings.normalization_method = norm
            s.feature_normalization_settings.normalization_time_s = float(rng.choice([1, 3, 30]))
            s.feature_normalization_settings.clip = float(rng.choice([0, 3]))
        try:
            s = NMSettings(**s.to_dict()).validate()
        except Exception:
            continue
        n_hops = int(min(rng.integers(6, 60), max(400 // n_ch, 4)))
        T = W + int(n_hops * sfreq / feat_hz) + int(rng.integers(0, 7))
        t = np.arange(T) / sfreq
        data = rng.standard_normal((n_ch, T)) * 10 + rng.uniform(-50, 50, (n_ch, 1))
        data += 8 * np.sin(2 * np.pi * rng.uniform(5, min(40, nyq / 3)) * t)[None]
        # a flat channel: -inf logs, empty extrema lists.  Not together with a normaliser: the reference's statistics of a
        # history that holds nan_to_num'ed infinities (+-DBL_MAX) are overflow artefacts of numpy's summation order
        # (mean -> -inf, std -> inf or NaN), and the scikit-learn transforms raise ValueError on an infinite feature
        if n_ch >= 3 and rng.random() < 0.25 and norm is None and not raw_norm:
            data[int(rng.integers(0, n_ch))] = 0.0
        return s, sfreq, data, 50, norm
    raise RuntimeError("no valid settings drawn")


def case_random_settings_wide(lib, seed):
    """`case_random_settings` over the wider generator; a drawn feature-normalisation method is checked as in
    case_pipeline_readme_default_zscore: the engine's normalised rows against the float64 oracle normaliser applied to
    the engine's OWN un-normalised rows."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.stream import Stream

    s, sfreq, data, line, norm = random_settings_wide(seed)
    ch = chmod.get_default_channels_from_data(data).to_dict("list")
    what = f"seed {seed} ({sfreq} Hz, {data.shape[0]} ch, {s.features.get_enabled()}, {s.preprocessing}, norm {norm})"
    try:
        df = Stream(sfreq, data=data, settings=s, line_noise=line, lib=lib).run(save_csv=False)
    except (ValueError, IndexError) as e:
        with pytest.raises(Exception):
            orc.run_stream(data, sfreq, s, ch, line_noise=line)
        return f"both raise: {e}"
    try:
        rows = orc.run_stream(data, sfreq, s, ch, line_noise=line)
    except IndexError as e:
        # data-dependent crash of the reference itself (sharpwaves.py:419-430: a window whose only trough has no peak
        # on one side still enters the steepness loop); a device kernel cannot raise, nothing to compare
        return f"reference raises on this recording: {e}"
    assert list(df.columns) == list(rows[0].keys()), f"{what}: columns differ"
    assert len(df) == len(rows)
    got = df.to_numpy(float)
    fs_win = sfreq
    starts, ends, _ = orc.window_schedule(data.shape[1], fs_win, s.sampling_rate_features_hz, s.segment_length_features_ms)
    W = int(ends[0] - starts[0])
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, W, line_noise=line, ends=ends)
    cols = list(df.columns)
    for i, r in enumerate(rows):
        want = np.array(list(r.values()))
        n_bad, rep, _ = parity.compare(cols[:-1], got[i, :-1], want[:-1], s, sfreq, 40.0, W, verifier=pv.row(i))
        assert n_bad == 0, f"{what} hop {i}\n{rep}"
        assert got[i, -1] == want[-1]
    if norm is not None:
        s_n = type(s)(**s.to_dict())
        s_n.postprocessing.feature_normalization = True
        got_n = Stream(sfreq, data=data, settings=s_n, line_noise=line, lib=lib).run(save_csv=False).to_numpy(float)
        fn = orc.FeatureNormalizer(s_n)
        sel = [i for i, k in enumerate(cols[:-1]) if "psd" not in k]   # normalize_psd = False (data_processor.py:283-290)
        want_n = got[:, :-1].copy()
        want_n[:, sel] = np.stack([fn.process(parity.widen_huge(r)) for r in got[:, sel]])

        def huge(a):   # nan_to_num'ed infinities: float64 max in the oracle, float32 max in the engine (tests/parity.py)
            return np.where(np.abs(a) >= parity.HUGE, np.sign(a) * np.inf, a)

        np.testing.assert_allclose(huge(got_n[:, :-1]), huge(want_n), rtol=1e-5, atol=2e-6, err_msg=what)


CHANNEL_TABLE_SEEDS = list(range(301, 321))


def random_channel_table(rng, n_ch):
    """A random channel table in the reference's layout: mixed types, bad / unused channels, a target channel, and every
    re-reference form of processing/rereference.py:33-86 ("average" inside the type, one named channel, "a&b", "None")."""
    import pandas as pd

    types = [str(rng.choice(["ecog", "ecog", "seeg", "lfp"])) for _ in range(n_ch)]
    names = [f"{t.upper()}_{'LR'[int(rng.integers(0, 2))]}_{i}" for i, t in enumerate(types)]
    status = ["bad" if rng.random() < 0.12 else "good" for _ in range(n_ch)]
    used = [0 if rng.random() < 0.12 else 1 for _ in range(n_ch)]
    target = [0] * n_ch
    if n_ch >= 4 and rng.random() < 0.5:
        t = int(rng.integers(0, n_ch))
        target[t], used[t], names[t], types[t] = 1, 0, "MOV_RIGHT", "misc"
    if not any(u == 1 and s == "good" and not tg for u, s, tg in zip(used, status, target)):
        used[0], status[0], target[0] = 1, "good", 0
    refs = []
    for i in range(n_ch):
        others = [names[j] for j in range(n_ch) if j != i and used[j] == 1 and not target[j]]
        mates = [j for j in range(n_ch) if j != i and used[j] == 1 and status[j] == "good" and types[j] == types[i]]
        u = rng.random()
        if target[i] or u < 0.2 or not others:
            refs.append("None")
        elif u < 0.6 and mates:
            refs.append("average")
        elif u < 0.85 or len(others) < 2:
            refs.append(str(rng.choice(others)))
        else:
            a, b = rng.choice(len(others), size=2, replace=False)
            refs.append(f"{others[a]}&{others[b]}")
    new = [n if r == "None" else (f"{n}-avgref" if r == "average" else f"{n}-{r}") for n, r in zip(names, refs)]
    return pd.DataFrame({"name": names, "rereference": refs, "used": used, "target": target, "type": types,
                         "status": status, "new_name": new})


def case_random_channel_tables(lib, seed):
    """Random channel tables (types, bad / unused channels, a target column, all re-reference forms) under random
    settings with re-referencing on: `Stream.run` against the oracle, columns (target column included) in order."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream

    s, sfreq, data, line = random_settings(seed)
    rng = np.random.default_rng(seed + 7)
    n_ch = int(rng.integers(3, 24))
    T = data.shape[1]
    data = rng.standard_normal((n_ch, T)) * 10 + rng.uniform(-50, 50, (n_ch, 1)) + data[:1]
    if "re_referencing" not in s.preprocessing:
        s.preprocessing = list(s.preprocessing) + ["re_referencing"]
        s = type(s)(**s.to_dict()).validate()
    tab = random_channel_table(rng, n_ch)
    ch = tab.to_dict("list")
    try:
        df = Stream(sfreq, channels=tab, data=data, settings=s, line_noise=line, lib=lib).run(data, save_csv=False)
    except (ValueError, IndexError) as e:
        with pytest.raises(Exception):
            orc.run_stream(data, sfreq, s, ch, line_noise=line)
        return f"both raise: {e}"
    rows = orc.run_stream(data, sfreq, s, ch, line_noise=line)
    assert list(df.columns) == list(rows[0].keys()), f"seed {seed}: columns differ\n{list(df.columns)[:6]}\n{list(rows[0])[:6]}"
    assert len(df) == len(rows)
    got = df.to_numpy(float)
    starts, ends, _ = orc.window_schedule(T, sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    W = int(ends[0] - starts[0])
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, W, line_noise=line, ends=ends)
    cols = list(df.columns)
    n_t = sum(tab["target"])
    feat_cols = len(cols) - 1 - n_t
    for i, r in enumerate(rows):
        want = np.array(list(r.values()))
        n_bad, rep, _ = parity.compare(cols[:feat_cols], got[i, :feat_cols], want[:feat_cols], s, sfreq, 40.0, W,
                                       verifier=pv.row(i))
        assert n_bad == 0, f"seed {seed} ({sfreq} Hz, {n_ch} ch)\n{tab}\nhop {i}\n{rep}"
        np.testing.assert_array_equal(got[i, feat_cols:], want[feat_cols:])   # time and target columns: exact


BURST_STREAM_SEEDS = list(range(401, 417))


def case_random_burst_streams(lib, seed):
    """Long streams through the burst detector only: the history fills, the walk switches from the workgroup kernel to
    the one-wave kernel in the middle of a batch (nmx_engine.inc, k_fill), and runs many hops in steady state -- over
    random sampling rates, feature rates, history lengths, percentiles and window lengths, fed in random batch sizes
    (state carried across `process_batch` calls) against the oracle's hop-by-hop `Bursts`."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    rng = np.random.default_rng(seed)
    sfreq = float(rng.choice([250, 500, 1000, 1000, 2000, 4000]))
    feat_hz = int(rng.choice([5, 10, 10, 20, 50]))
    seg_ms = int(rng.choice([250, 500, 1000]))
    hop = int(sfreq / feat_hz)
    W = int(sfreq * seg_ms / 1000)
    dur = float(rng.choice([0.5, 1, 2, 5]))
    base = NMSettings.get_default().to_dict()
    pool = [("alpha", [8, 12]), ("low_beta", [13, 20]), ("high_beta", [20, 35]), ("low_gamma", [60, 80])]
    keep = [b for b in pool if b[1][1] + 10 < sfreq / 2]
    keep = [keep[i] for i in sorted(rng.choice(len(keep), size=int(rng.integers(1, len(keep) + 1)), replace=False))]
    base["frequency_ranges_hz"] = {n: r for n, r in keep}
    s = NMSettings(**base)
    s.sampling_rate_features_hz = feat_hz
    s.segment_length_features_ms = seg_ms
    s.features.disable_all()
    s.features.bursts = True
    s.bursts_settings.frequency_bands = [n for n, _ in keep]
    s.bursts_settings.time_duration_s = dur
    s.bursts_settings.threshold = float(rng.choice([50, 75, 75, 90, 97]))
    s.bandpass_filter_settings.segment_lengths_ms = {n: seg_ms for n, _ in keep}
    s.preprocessing = []
    s.postprocessing.feature_normalization = False
    s = NMSettings(**s.to_dict()).validate()
    C = int(rng.integers(1, 6))
    n_fill = int(np.ceil(max(dur * sfreq - W, 0) / hop)) + 1      # hops until the history is full
    n_hops = int(min(n_fill + rng.integers(20, 120), 600))
    T = W + (n_hops - 1) * hop
    t = np.arange(T) / sfreq
    x = rng.standard_normal((C, T)) * 10 + 6 * np.sin(2 * np.pi * 17 * t) * (1 + np.sin(2 * np.pi * 0.7 * t))
    ch = [f"c{i}" for i in range(C)]
    eng = HotPathEngine(s, ch, sfreq, lib=lib)
    ob = orc.Bursts(s, ch, sfreq)
    starts = np.arange(n_hops, dtype=np.int64) * hop
    got, a = [], 0
    while a < n_hops:                     # random batch sizes; the first one often ends inside the fill phase
        b = min(n_hops, a + int(rng.integers(1, max(n_fill, 2) * 2)))
        got.append(eng.process_batch(x, starts[a:b]))
        a = b
    got = np.concatenate(got)
    what = f"seed {seed}: {sfreq} Hz, hop {hop}, W {W}, history {dur} s, q {s.bursts_settings.threshold}, {C} ch, {n_hops} hops"
    for i, st in enumerate(starts):
        w = x[:, st:st + W]
        want = ob.calc_feature(w)
        assert list(want) == eng.keys
        ver = parity.Verifier(s, ch, sfreq, w, bursts=ob)
        n_bad, rep, _ = parity.compare(eng.keys, got[i], list(want.values()), s, sfreq, 40.0, W, verifier=ver)
        assert n_bad == 0, f"{what}; hop {i} (history full from hop {n_fill})\n{rep}"
    eng.close()


def case_random_window_by_window(lib, seed, wide=False):
    """The reference's real-time loop hands ONE window per call to `DataProcessor.process`; the batch driver hands all
    hops of a recording to the library at once.  Both are the same kernels and the same carried state (burst history,
    Kalman filters, normalisers), so for a random settings point the rows must agree to the last bit of every feature
    and to one fp32 ulp behind the feature normaliser (switched on here: its history is part of the state)."""
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.data_processor import DataProcessor
    from py_neuromodulation_amd.stream import Stream

    if wide:   # (+ resampling, pre-processing filters, raw normalisation, Kalman, every normaliser method)
        s, sfreq, data, line, _ = random_settings_wide(seed)
    else:
        s, sfreq, data, line = random_settings(seed)
    s = type(s)(**s.to_dict())
    s.postprocessing.feature_normalization = True
    if not wide:
        s.feature_normalization_settings.normalization_time_s = 1.0
    data = data.astype(np.float32).astype(np.float64)
    try:
        st = Stream(sfreq, data=data, settings=s, line_noise=line, lib=lib)
        df = st.run(save_csv=False)
    except (ValueError, IndexError, NotImplementedError) as e:
        return f"not runnable: {e}"
    from oracle import nm_oracle as orc   # (the schedule only)

    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    if len({int(b - a) for a, b in zip(starts, ends)}) != 1:
        return "ragged window lengths: one processor per length in the batch driver"
    dp = DataProcessor(sfreq=sfreq, settings=s, channels=chmod.get_default_channels_from_data(data), line_noise=line,
                       lib=lib)
    cols = [c for c in df.columns if c != "time"]
    got = df[cols].to_numpy(float)
    for i, (a, b) in enumerate(zip(starts, ends)):
        row = dp.process(data[:, a:b])
        assert list(row) == cols
        a32, b32 = np.array(list(row.values()), np.float32), got[i].astype(np.float32)
        assert np.array_equal(np.isnan(a32), np.isnan(b32)), f"seed {seed} hop {i}: NaN pattern"
        ok = ~np.isnan(a32)
        # features: identical.  After the normaliser: within one fp32 ulp -- its float64 sums are rebuilt from the history
        # at the start of every library call and slide inside a call, so the two call shapes add in a different order
        np.testing.assert_array_max_ulp(a32[ok], b32[ok], maxulp=1)


# ---- user-registered NMFeature plugins (features/feature_processor.py:52-53,90-108) -------------------------
def case_user_features(lib, tags=("raw", "two", "ex")):
    """The reference's own Stream.run with plugins registered through add_custom_feature wrote
    tests/golden/user_features.npz; the fused orchestrator must return the same table: the plugin columns after
    the built-in ones in registration order, normalised with them ("psd" keys skipped), NaN policy by substring.

    raw: no pre-processing, no normaliser -- the plugins see the float64 window itself: 1e-12.
    two / ex: re-reference (+ notch for ex) and a z-score.  A z-score amplifies fp32 rounding by value / spread, so
      the composition is checked stage by stage like case_pipeline_readme_default_zscore: (1) the un-normalised
      plugin columns of the same run against the plugins applied to the ORACLE's float64 pre-processed windows
      (1e-5 relative + 2e-6 x the window's amplitude: a mean is conditioned by the amplitude of what it averages),
      (2) the normalised table against the float64 oracle normaliser on the engine's own rows, (3) the reference's
      normalised table as a sanity bound."""
    import json

    import py_neuromodulation_amd as nmx
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests import user_plugins as up
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("user_features")
    cases = {"ex": ({"channel_mean": up.ChannelMean}, "ex_data", "ex_channels_json", 10),
             "two": ({"channel_mean": up.ChannelMean, "hop_stats": up.HopStats}, "two_data", "two_channels_json", None),
             "raw": ({"channel_mean": up.ChannelMean, "hop_stats": up.HopStats}, "two_data", "two_channels_json", None)}
    for tag in tags:
        plugins, data_key, ch_key, feat_hz = cases[tag]
        for name, cls in plugins.items():
            nmx.add_custom_feature(name, cls)
        try:
            s = settings_from_json(g[f"{tag}_settings_json"])
            data = g[data_key]
            ch = json.loads(str(g[ch_key]))
            cols = [str(c) for c in g[f"{tag}_columns"]]
            want = g[f"{tag}_values"]

            def run(settings, x=None):
                st = Stream(sfreq=1000.0, channels=ch, settings=settings, line_noise=50, lib=lib)
                return st, st.run(data if x is None else x, save_csv=False)

            st, df = run(s)
            assert list(df.columns) == cols, f"{tag}: columns / order differ from the reference"
            got = df.to_numpy(dtype=np.float64)
            assert got.shape == want.shape
            assert np.array_equal(np.isnan(got), np.isnan(want)), f"{tag}: NaN policy"
            np.testing.assert_array_equal(got[:, -1], want[:, -1])   # time
            n_builtin = len(st.data_processor.engine.keys)
            user_cols = cols[n_builtin:-1]
            assert user_cols == list(st.data_processor.user_keys)
            starts, ends, _ = orc.window_schedule(data.shape[1], 1000.0, s.sampling_rate_features_hz,
                                                  s.segment_length_features_ms)
            if tag == "raw":
                pv = parity.PipelineVerifiers(s, ch, 1000.0, data, starts, 1000, line_noise=50)
                for r in range(len(got)):
                    n_bad, rep, _ = parity.compare(cols[:n_builtin], got[r, :n_builtin], want[r, :n_builtin], s, 1000.0,
                                                   500.0, 1000, verifier=pv.row(r))
                    assert n_bad == 0, f"raw row {r}\n{rep}"
                np.testing.assert_allclose(got[:, n_builtin:-1], want[:, n_builtin:-1], rtol=1e-12, atol=0, equal_nan=True)
                continue
            # (1) un-normalised plugin columns vs the plugins on the oracle's float64 pre-processed windows
            s_raw = type(s)(**s.to_dict())
            s_raw.postprocessing.feature_normalization = False
            # (features are computed from nan_to_num(window) and normalised BEFORE the NaN policy blanks them,
            # stream/data_processor.py:255,263-306: the history of the normaliser holds the values of the cleaned data)
            st_raw, df_raw = run(s_raw, np.nan_to_num(data))
            raw = df_raw.to_numpy(dtype=np.float64)[:, :-1]
            dp = orc.DataProcessor(1000.0, s_raw, ch, 50)
            insts = [cls(s_raw, dp.ch_names_used, dp.sfreq) for cls in plugins.values()]
            for r, (a, b) in enumerate(zip(starts, ends)):
                w = dp.preprocess(np.nan_to_num(data[:, a:b])[dp.feature_idx])
                d: dict = {}
                for f in insts:
                    d.update(f.calc_feature(w))
                assert list(d) == user_cols
                amp = float(np.abs(w).max())
                ok = ~np.isnan(raw[r, n_builtin:])   # NaN policy already compared above
                np.testing.assert_allclose(raw[r, n_builtin:][ok], np.array(list(d.values()))[ok], rtol=1e-5,
                                           atol=2e-6 * amp, err_msg=f"{tag} row {r}")
            # (2) the normaliser on the engine's own rows ("psd" keys pass through, stream/data_processor.py:263-290)
            keys = cols[:-1]
            non_psd = [i for i, k in enumerate(keys) if "psd" not in k]
            norm = orc.FeatureNormalizer(s)
            # the plugin columns enter the device normaliser as float32, like the built-in ones: same inputs on both sides
            raw[:, n_builtin:] = raw[:, n_builtin:].astype(np.float32)
            want_n = raw.copy()
            for r in range(len(raw)):
                want_n[r, non_psd] = norm.process(raw[r, non_psd].copy())
            fin = ~np.isnan(got[:, :-1])
            np.testing.assert_allclose(got[:, :-1][fin], want_n[fin], rtol=1e-5, atol=2e-6, err_msg=tag)
            # (3) the reference's normalised table
            err = np.abs(got[1:, :-1] - want[1:, :-1])
            assert np.nanmedian(err) < 1e-4, tag
            assert np.nanmedian(err[:, n_builtin:]) < 1e-4, tag
        finally:
            for name in plugins:
                nmx.remove_custom_feature(name)


def case_ragged_bursts(lib):
    """State that crosses window LENGTHS (reference golden tests/golden/ragged_bursts.npz, sfreq = 1111.111 Hz: windows of
    1111 and 1112 samples): one burst history and one set of Kalman filters for the whole stream -- the per-length plans
    of the batch driver hand the state blob over where the length changes (stream.py)."""
    import json

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("ragged_bursts")
    s = settings_from_json(g["settings_json"])
    sfreq, data = float(g["sfreq"]), g["data"]
    ch = json.loads(str(g["channels_json"]))
    df = Stream(sfreq, channels=ch, settings=s, line_noise=50, lib=lib).run(data, save_csv=False)
    cols = [str(c) for c in g["columns"]]
    assert list(df.columns) == cols
    got, want = df.to_numpy(float), g["values"]
    assert got.shape == want.shape
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    assert sorted({int(b - a) for a, b in zip(starts, ends)}) == sorted(set(int(x) for x in g["window_lengths"]))
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, 1111, line_noise=50, ends=ends)
    for i in range(len(got)):
        n_bad, rep, _ = parity.compare(cols[:-1], got[i, :-1], want[i, :-1], s, sfreq, 60.0, 1111, verifier=pv.row(i))
        assert n_bad == 0, f"hop {i}\n{rep}"
    np.testing.assert_array_equal(got[:, -1], want[:, -1])


def case_long_windows(lib, tags=("sw_default", "sw_all", "rn_median", "rn_zscore_median", "rn_robust", "rn_minmax")):
    """Windows of 7000 samples (reference golden tests/golden/long_windows.npz, 7 kHz, 700-sample hops): sharp waves
    beyond 4092 samples (per-lane chunks of more than 64 samples take the two-pass extrema walk, the lists are sized by the
    window) and the order-statistic raw normalisers beyond window + hop = 6484 samples (their merge lists move from LDS
    to device memory)."""
    import json

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("long_windows")
    sfreq, data = float(g["sfreq"]), g["data"].astype(np.float64)
    for tag in tags:
        s = settings_from_json(g[f"{tag}_settings_json"])
        ch = json.loads(str(g[f"{tag}_channels_json"]))
        x = data + 20.0 if tag.startswith("rn_") else data
        df = Stream(sfreq, channels=ch, settings=s, line_noise=50, lib=lib).run(x, save_csv=False)
        cols = [str(c) for c in g[f"{tag}_columns"]]
        assert list(df.columns) == cols, tag
        got, want = df.to_numpy(float), g[f"{tag}_values"]
        assert got.shape == want.shape, tag
        np.testing.assert_array_equal(got[:, -1], want[:, -1])
        starts, ends, _ = orc.window_schedule(x.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
        if tag.startswith("rn_"):
            # z-scores of fp32 samples against float64 order statistics (case_raw_normalizer's tolerance); the fft
            # columns are log10 of band means of the normalised window
            raw = [j for j, c in enumerate(cols[:-1]) if c.endswith("_raw")]
            oth = [j for j, c in enumerate(cols[:-1]) if not c.endswith("_raw")]
            np.testing.assert_allclose(got[:, raw], want[:, raw], rtol=2e-5, atol=5e-6, err_msg=tag)
            np.testing.assert_allclose(got[:, oth], want[:, oth], rtol=2e-5, atol=2e-5, err_msg=tag)
            continue
        pv = parity.PipelineVerifiers(s, ch, sfreq, x, starts, 7000, line_noise=50, ends=ends)
        for i in range(len(got)):
            n_bad, rep, _ = parity.compare(cols[:-1], got[i, :-1], want[i, :-1], s, sfreq, 20.0, 7000, verifier=pv.row(i))
            assert n_bad == 0, f"{tag} hop {i}\n{rep}"


def case_ragged_rawnorm(lib):
    """raw_normalization with ragged window lengths (reference golden tests/golden/ragged_rawnorm.npz, 1111.111 Hz): one
    sample history for the whole stream -- the state blob of the raw normaliser carries its ring capacity, the plan of
    the other window length re-lays the histories into its own rings (nmx_state_import)."""
    import json

    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("ragged_rawnorm")
    sfreq, data = float(g["sfreq"]), g["data"]
    for tag in ("zscore", "median"):
        s = settings_from_json(g[f"{tag}_settings_json"])
        ch = json.loads(str(g[f"{tag}_channels_json"]))
        df = Stream(sfreq, channels=ch, settings=s, line_noise=50, lib=lib).run(data, save_csv=False)
        cols = [str(c) for c in g[f"{tag}_columns"]]
        assert list(df.columns) == cols, tag
        got, want = df.to_numpy(float), g[f"{tag}_values"]
        assert got.shape == want.shape, tag
        np.testing.assert_array_equal(got[:, -1], want[:, -1])
        # z-scores of fp32 samples against float64 statistics (case_raw_normalizer's tolerance); fft columns are log10
        # band means of the normalised window
        raw = [j for j, c in enumerate(cols[:-1]) if c.endswith("_raw")]
        oth = [j for j, c in enumerate(cols[:-1]) if not c.endswith("_raw")]
        np.testing.assert_allclose(got[:, raw], want[:, raw], rtol=2e-5, atol=5e-6, err_msg=tag)
        np.testing.assert_allclose(got[:, oth], want[:, oth], rtol=5e-5, atol=2e-5, err_msg=tag)


# ---- input layouts: the reference takes whatever NumPy / pandas hand it (stream/stream.py:95-108: a DataFrame
# becomes ``to_numpy().transpose()``, a Fortran-ordered view) ---------------------------------------------------
def _layout_variants(base_tc: np.ndarray):
    """``base_tc`` is (samples, channels), C-ordered.  -> {name: (channels, samples) array of the SAME values}."""
    C_ord = np.ascontiguousarray(base_tc.T)
    T, C = base_tc.shape
    wide = np.empty((C, 2 * T), base_tc.dtype)
    wide[:, ::2] = C_ord
    wide[:, 1::2] = -7.0
    tall = np.full((2 * C, T + 5), 3.0, base_tc.dtype)
    tall[::2, :T] = C_ord
    return {"c_order": C_ord, "transposed_view": base_tc.T, "strided_columns": wide[:, ::2],
            "strided_rows": tall[::2, :T]}


def case_input_layouts(lib, large: bool = False):
    """{C order, ``.T`` view of a (samples, channels) array, strided columns, strided rows, pd.DataFrame} x {float32,
    float64} x {no offset, one row 1e5 spreads off zero} through `Stream.run`, `DataProcessor.process_batch`,
    `HotPathEngine.process_batch` and `DataProcessor.process` (one window): every layout returns what the C-ordered
    array returns, bit for bit.  ``large``: T x C above the 2**18 elements from which the cast runs on the staging
    threads (the small sizes take the in-line conversions)."""
    import pandas as pd

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.data_processor import DataProcessor
    from py_neuromodulation_amd.engine import HotPathEngine
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.raw_hjorth = s.features.fft = s.features.linelength = s.features.return_raw = True
    s.postprocessing.feature_normalization = False
    C, T = 6, (48000 if large else 3000)
    assert (C * T >= (1 << 18)) == large
    rng = np.random.default_rng(77)
    for dtype in (np.float64, np.float32):
        for level in (0.0, 1e5):
            base = rng.standard_normal((T, C)) * 50
            base[:, 2] += level * 50
            base = base.astype(dtype)
            variants = _layout_variants(base)
            want_df = None
            channels = chmod.get_default_channels_from_data(variants["c_order"])
            names = channels["name"].to_list()
            for name, arr in list(variants.items()) + [("dataframe", pd.DataFrame(base, columns=names))]:
                what = f"{np.dtype(dtype).name}, level {level:g}, {name}"
                if name != "dataframe":
                    np.testing.assert_array_equal(arr, variants["c_order"])
                df = Stream(sfreq=1000.0, data=arr, channels=channels, settings=s, line_noise=50, lib=lib).run(save_csv=False)
                if want_df is None:
                    want_df = df
                    assert np.isfinite(df.to_numpy()).all() and len(df) == (T - 1000) // 100 + 1
                else:
                    assert list(df.columns) == list(want_df.columns)
                    np.testing.assert_array_equal(df.to_numpy(), want_df.to_numpy(), err_msg="Stream.run: " + what)
                if name == "dataframe":
                    continue
                if not large:   # the same recording on two plans (sharding.MultiDeviceProcessor stages the rows itself)
                    two = Stream(sfreq=1000.0, data=arr, channels=channels, settings=s, line_noise=50, lib=lib,
                                 devices=[0, 0]).run(save_csv=False).to_numpy()
                    if name == "c_order":
                        want_two = two
                    else:
                        np.testing.assert_array_equal(two, want_two, err_msg="Stream(devices=[0, 0]).run: " + what)
                starts = np.arange(0, T - 1000 + 1, 100)[:12]
                dp = DataProcessor(sfreq=1000.0, settings=s, channels=channels, line_noise=50, lib=lib)
                rows = dp.process_batch(arr, starts)
                np.testing.assert_array_equal(rows[:, :len(dp.keys)], want_df.to_numpy()[:12, :len(dp.keys)],
                                              err_msg="DataProcessor.process_batch: " + what)
                dp.reset()
                one = dp.process(arr[:, :1000])
                if name == "c_order":
                    want_one = one
                else:
                    assert one == want_one, "DataProcessor.process: " + what
                eng = HotPathEngine(s, names, 1000.0, lib=lib)
                o = eng.process_batch(arr, starts)
                eng.close()
                if name == "c_order":
                    want_o = o
                else:
                    np.testing.assert_array_equal(o, want_o, err_msg="HotPathEngine.process_batch: " + what)


def case_input_layouts_single_channel(lib):
    """One channel: NumPy reports an arbitrary row stride for a (1, T) array (the transpose of a (T, 1) column has a
    row "pitch" of one element) -- the reference's real-time demo streams exactly one channel
    (examples/plot_6_real_time_demo.py:54-106)."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.raw_hjorth = s.features.fft = True
    s.postprocessing.feature_normalization = False
    s.preprocessing = ["notch_filter"]
    rng = np.random.default_rng(78)
    for dtype in (np.float64, np.float32):
        for level in (0.0, 1e7):
            col = (rng.standard_normal((2500, 1)) * 50 + level).astype(dtype)
            a = Stream(sfreq=1000.0, data=np.ascontiguousarray(col.T), settings=s, line_noise=50, lib=lib).run(save_csv=False)
            b = Stream(sfreq=1000.0, data=col.T, settings=s, line_noise=50, lib=lib).run(save_csv=False)
            np.testing.assert_array_equal(a.to_numpy(), b.to_numpy())


# ---- the recording the reference's own tests run on (tests/conftest.py:8-69) --------------------------------------
def real_recording():
    """-> (golden, data [10, 19001] float64 in volt, channel table dict, sfreq): tests/golden/real_recording.npz holds the
    samples as the reference's BrainVision file stores them (float32, multiplexed) and MNE's scale per channel."""
    import json

    from tests.helpers import load_golden

    g = load_golden("real_recording")
    data = g["stored"].T.astype(np.float64) * g["scale"][:, None]
    return g, data, json.loads(str(g["channels_json"])), float(g["sfreq"])


def case_real_recording(lib, devices=None):
    """`Stream.run` on sub-testsub's iEEG run with the channel table of the reference's fixtures (ECoG against the common
    average, LFP contacts bipolar, MOV_RIGHT the target), the default pre-processing and all nine feature families at
    10 Hz: 181 hops x 353 columns against the reference's own DataFrame under the standard policy (tests/parity.py);
    then the default z-score on top, checked as a composition (case_pipeline_readme_default_zscore)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import settings_from_json

    g, data, ch, sfreq = real_recording()
    s = settings_from_json(g["nonorm_settings_json"])
    kw = {"devices": list(devices)} if devices else {}
    df = Stream(sfreq=sfreq, channels=ch, settings=s, line_noise=50, lib=lib, **kw).run(data, save_csv=False)
    cols = [str(c) for c in g["nonorm_columns"]]
    assert list(df.columns) == cols, "DataFrame columns / order differ from the reference"
    got, want = df.to_numpy(dtype=np.float64), g["nonorm_values"]
    assert got.shape == want.shape == (181, 353)
    nf = cols.index("time")
    np.testing.assert_array_equal(got[:, nf:], want[:, nf:])   # time and the target channel's samples: exact
    starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
    pv = parity.PipelineVerifiers(s, ch, sfreq, data, starts, 1000, line_noise=50)
    used = [i for i, (u, t) in enumerate(zip(ch["used"], ch["target"])) if u == 1 and t == 0]
    amp = float(np.abs(data[used] - data[used].mean(axis=1, keepdims=True)).max())
    for r in range(len(got)):
        n_bad, rep, _ = parity.compare(cols[:nf], got[r, :nf], want[r, :nf], s, sfreq, amp, 1000, verifier=pv.row(r))
        assert n_bad == 0, f"row {r}\n{rep}"
    # the default z-score (default_settings.yaml:69-78) inside the plan: the engine's normalised rows == the float64 oracle
    # normaliser applied to the engine's OWN un-normalised rows (a z-score amplifies fp32 rounding by value / spread)
    sz = settings_from_json(g["zscore_settings_json"])
    dz = Stream(sfreq=sfreq, channels=ch, settings=sz, line_noise=50, lib=lib, **kw).run(data, save_csv=False)
    assert list(dz.columns) == [str(c) for c in g["zscore_columns"]] == cols
    gz, wz = dz.to_numpy(dtype=np.float64), g["zscore_values"]
    norm = orc.FeatureNormalizer(sz)
    want_n = np.stack([norm.process(r.copy()) for r in got[:, :nf]])
    np.testing.assert_array_equal(gz[0, :nf], got[0, :nf])
    np.testing.assert_allclose(gz[:, :nf], want_n, rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(gz[:, nf:], wz[:, nf:])
    assert np.nanmedian(np.abs(gz[1:, :nf] - wz[1:, :nf])) < 1e-4


# ---- several members of one re-reference group on the rail in the same sample ---------------------------------------
def _rail_class(v, key, settings, ours: bool):
    """0: an ordinary value; +-1: derived from samples on the rail, with that sign; 2: NaN; 3: a spectral entry of a window
    with samples on the rail.  The reference's rail is +-DBL_MAX (np.nan_to_num of +-inf in float64), the engine's
    +-FLT_MAX: features of such a window are "huge" on either scale -- beyond 1e200 there, beyond 1e25 (or +-inf: the
    square of 1e38 is not a float32) here.  A transform of such a window adds terms of the size of the rail: whether a bin
    comes out as 300-odd (log10 of a finite sum), +inf or NaN (inf - inf in a butterfly) is the summation order of the FFT
    at hand, in pocketfft as in the engine's -- one class."""
    fam = parity.family_of(key)
    if fam in ("fft", "stft", "welch"):
        # (log10 values of this recording's ordinary bins lie below 4; an STFT / Welch entry averages a few segments, one of
        # them on the rail: 300 / 3 there, 38 / 3 here)
        logbig = (10.0 if ours else 60.0) if getattr(settings, fam + "_settings").log_transform else (1e25 if ours else 1e200)
        return 0 if (np.isfinite(v) and v < logbig) else 3
    if np.isnan(v):
        return 2
    if abs(v) >= (1e25 if ours else 1e200):
        return int(np.sign(v))
    return 0


def case_inf_members(lib, run=None, tag="", spectral_nan_ok=True):
    """tests/golden/inf_members.npz (the reference's own run): two and three members of the common-average group at +inf
    in one sample, +inf and -inf together, one member at -inf twice.  Every entry must fall in the reference's class
    (ordinary / rail-derived with the same sign / NaN) and ordinary entries meet the stated tolerances -- on one plan, on
    two (``Stream(devices=[0, 0])``: the group sums come from the host as saturating hi / lo pairs) and, under gloo, on
    two ranks (tests/test_sharding_gloo.py passes its own ``run``)."""
    import json

    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("inf_members")
    s = settings_from_json(g[tag + "settings_json"])
    ch = json.loads(str(g["channels_json"]))
    cols = [str(c) for c in g[tag + "columns"]]
    want = g[tag + "values"]
    runs = {}
    if run is not None:
        runs["caller"] = run
    else:
        runs["one plan"] = lambda: Stream(sfreq=1000.0, channels=ch, settings=s, lib=lib).run(g["data"], save_csv=False)
        runs["two plans"] = lambda: Stream(sfreq=1000.0, channels=ch, settings=s, lib=lib, devices=[0, 0]).run(
            g["data"], save_csv=False)
    patterns = {}
    for name, fn in runs.items():
        df = fn()
        assert list(df.columns) == cols
        got = df.to_numpy(dtype=np.float64)
        assert got.shape == want.shape
        cls_g = np.array([[_rail_class(got[r, c], cols[c], s, True) for c in range(len(cols))] for r in range(len(got))])
        cls_w = np.array([[_rail_class(want[r, c], cols[c], s, False) for c in range(len(cols))] for r in range(len(got))])
        # +inf and -inf in one sample (2650): on every OTHER channel the two terms cancel and the exact value is an ordinary
        # one; the reference returns the rounding residue of DBL_MAX / 7 - DBL_MAX / 7 (1e-20 of the rail: LineLength
        # 4e285, Activity DBL_MAX), the engine's float64 sum cancels exactly.  Nothing to compare on those windows.
        starts = np.arange(0, g["data"].shape[1] - 1000 + 1, 100)
        residue = np.zeros_like(cls_w, dtype=bool)
        for r in np.flatnonzero((starts <= 2650) & (2650 < starts + 1000)):
            residue[r] = [not (c.startswith("ch2_") or c.startswith("ch5_")) and c != "time" for c in cols]
            # ... and the two members themselves overflow to +-inf (DBL_MAX + DBL_MAX / 7): whether a transform of a segment
            # with an infinite sample returns inf or NaN (inf - inf in a butterfly; np.nanmean then skips the segment
            # and the entry is an ORDINARY number) is pocketfft's summation order -- no class to hold the engine to
            residue[r] |= np.array([(c.startswith("ch2_") or c.startswith("ch5_"))
                                    and parity.family_of(c) in ("fft", "stft", "welch") for c in cols])
        cls_w[residue] = cls_g[residue]
        if not spectral_nan_ok:
            # the wave-level kernels of the MI355X library report an overflowed transform as +inf, never as NaN
            # (nmx_k_timeosc_w1000.h: NmxBandAcc::railed): where the reference holds a finite number, so does the engine, or +inf
            lost = np.argwhere(np.isnan(got) & np.isfinite(want) & ~residue)
            assert len(lost) == 0, f"{name}: NaN where the reference is finite: " + "; ".join(
                f"hop {r} {cols[c]}: want {want[r, c]!r}" for r, c in lost[:8])
        bad = np.argwhere(cls_g != cls_w)
        assert len(bad) == 0, f"{name}: {len(bad)} entries in another class than the reference's, e.g. " + "; ".join(
            f"hop {r} {cols[c]}: got {got[r, c]!r} want {want[r, c]!r}" for r, c in bad[:8])
        counts = {k: int((np.abs(cls_w) == k).sum()) for k in (1, 2, 3)}
        assert counts[1] > 100 and counts[3] > 400, f"the golden lost its rails: {counts}"
        nf = cols.index("time")
        for r in range(len(got)):
            keep = [c for c in range(nf) if cls_w[r, c] == 0 and not residue[r, c]]
            n_bad, rep, _ = parity.compare([cols[c] for c in keep], got[r, keep], want[r, keep], s, 1000.0, 200.0, 1000)
            assert n_bad == 0, f"{name}, hop {r}\n{rep}"
        patterns[name] = cls_g
    return patterns


# ---- drift is not an offset: the engine's split carries one CONSTANT per row (nmx_engine_dc.inc) ----------------------------
def case_trends(lib):
    """Recordings that wander: a 0.3 Hz swell of 10^3 spreads, one of 10^4, a linear drift that reaches 10^4 spreads over the
    recording -- next to two well-behaved rows.  A window of such a row holds values hundreds to thousands of times its
    signal, and a float32 sample rounds at the VALUE's size: what every feature then reads carries that rounding (a
    per-row linear term would have to travel through every kernel's sums like the constant does; it does not).  The
    policy's noise levels are those of what the samples hold (tests/parity.py: `Verifier._held`), so the misses are
    explained entry by entry -- this case COUNTS them, per row, so that the budget file shows what a trend costs:
    the quiet rows stay clean (at most an ordinary near-null STFT bin), the wandering rows' smooth time-domain features stay within 1e-3."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    s.features.disable_all()
    for f in ("raw_hjorth", "fft", "welch", "stft", "linelength", "return_raw"):
        setattr(s.features, f, True)
    s.preprocessing = []
    s.postprocessing.feature_normalization = False
    s.sampling_rate_features_hz = 5
    T, sigma = 5000, 50.0
    rng = np.random.default_rng(91)
    t = np.arange(T) / 1000.0
    data = rng.standard_normal((5, T)) * sigma + 8 * np.sin(2 * np.pi * 21 * t)
    data[1] += 1e3 * sigma * np.sin(2 * np.pi * 0.3 * t + 0.4)
    data[2] += 1e4 * sigma * np.sin(2 * np.pi * 0.3 * t + 2.0)
    data[3] += 1e4 * sigma * (t / t[-1])
    data = data.astype(np.float32).astype(np.float64)   # identical windows on both sides
    ch = {"name": [f"ch{i}" for i in range(5)], "rereference": ["None"] * 5, "used": [1] * 5, "target": [0] * 5,
          "type": ["ecog"] * 5, "status": ["good"] * 5, "new_name": [f"ch{i}" for i in range(5)]}
    df = Stream(1000.0, channels=ch, settings=s, line_noise=50, lib=lib).run(data, save_csv=False)
    rows = orc.run_stream(data, 1000.0, s, ch)
    cols = list(df.columns)
    assert cols == list(rows[0].keys())
    got = df.to_numpy(float)
    starts, ends, _ = orc.window_schedule(T, 1000.0, s.sampling_rate_features_hz, s.segment_length_features_ms)
    pv = parity.PipelineVerifiers(s, ch, 1000.0, data, starts, 1000, line_noise=50)
    before = dict(parity.STATS["forgiven"])
    quiet = [i for i, c in enumerate(cols) if c.startswith("ch0_") or c.startswith("ch4_")]
    n_quiet = 0   # (an ordinary near-null STFT bin now and then: 1 of 476 entries on the MI355X)
    for r in range(len(rows)):
        want = np.array([rows[r][c] for c in cols])
        q0 = sum(parity.STATS["forgiven"].values())
        n_bad, rep, _ = parity.compare([cols[i] for i in quiet], got[r, quiet], want[quiet], s, 1000.0, 4 * sigma, 1000,
                                       verifier=pv.row(r))
        assert n_bad == 0, f"quiet rows, hop {r}\n{rep}"
        n_quiet += sum(parity.STATS["forgiven"].values()) - q0
        wander = [i for i in range(len(cols) - 1) if i not in quiet]
        n_bad, rep, _ = parity.compare([cols[i] for i in wander], got[r, wander], want[wander], s, 1000.0, 1e4 * sigma, 1000,
                                       verifier=pv.row(r))
        assert n_bad == 0, f"wandering rows, hop {r}\n{rep}"
        smooth = [i for i in wander if parity.family_of(cols[i]) in ("hjorth", "linelength")]
        np.testing.assert_allclose(got[r, smooth], want[smooth], rtol=1e-3)
    assert n_quiet <= 3, f"{n_quiet} misses on the rows without a trend"
    return {k: v - before.get(k, 0) for k, v in parity.STATS["forgiven"].items()}


def case_dc_nan(lib):
    """tests/golden/dc_nan.npz (the reference's own run): NaN samples on a channel 10^5 spreads off zero with NO re-reference in
    front, behind a notch and without any pre-processing.  The float64 recording is split on the host (`engine._host_offsets`),
    so a NaN reaches the device as NaN in the split domain and has to become the recording's value 0 there -- minus the
    row's constant -- or the burst history of that channel parts from the reference's for the rest of the stream."""
    import json

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.stream import Stream
    from tests.helpers import load_golden, settings_from_json

    g = load_golden("dc_nan")
    data = g["data"]
    ch = json.loads(str(g["channels_json"]))
    for tag in ("notch", "nopre"):
        s = settings_from_json(g[f"{tag}_settings_json"])
        st = Stream(sfreq=1000.0, channels=ch, settings=s, line_noise=50, lib=lib)
        df = st.run(data, save_csv=False)
        assert st.data_processor.engine._dc_any, "the host split did not engage"
        cols = [str(c) for c in g[f"{tag}_columns"]]
        assert list(df.columns) == cols
        got, want = df.to_numpy(dtype=np.float64), g[f"{tag}_values"]
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).any()
        starts, ends, _ = orc.window_schedule(data.shape[1], 1000.0, s.sampling_rate_features_hz, s.segment_length_features_ms)
        pv = parity.PipelineVerifiers(s, ch, 1000.0, data, starts, 1000, line_noise=50)
        for r in range(len(got)):
            ok = ~np.isnan(want[r])
            keys = [k for k, o in zip(cols, ok) if o]
            n_bad, rep, _ = parity.compare(keys, got[r][ok], want[r][ok], s, 1000.0, 80.0, 1000, verifier=pv.row(r))
            assert n_bad == 0, f"{tag}, hop {r}\n{rep}"


class _default_library:
    """The plugin classes load the package's own library; the cases that drive them run on `lib` through this."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        from py_neuromodulation_amd import _lib

        self.prev, _lib._default = _lib._default, self.lib

    def __exit__(self, *exc):
        from py_neuromodulation_amd import _lib

        _lib._default = self.prev


def case_standalone_classes_any_length(lib):
    """What the reference's own tests do with the stand-alone classes (found by running /root/reference/tests with the
    classes swapped, tests/golden/run_reference_tests.py): MNEFilter.filter_data on 10 s at 4 kHz
    (tests/test_nm_filter.py:11-96), NotchFilter.process on a 1-D window of `sfreq` samples at 150 / 200 Hz
    (tests/test_notch_filter.py:7-31), a recording longer than one plan's window through the notch and the
    preprocessing filter -- segments with halos (engine.long_segments) must equal ONE long convolution."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings, features, fir_design
    from py_neuromodulation_amd.engine import MAX_PLAN_WINDOW, long_segments
    from py_neuromodulation_amd.processing import NotchFilter, PreprocessingFilter

    # every output sample exactly once, each at least `halo` from a cut that is not an end of the recording
    for T, halo in ((40000, 1999), (16385, 499), (100000, 7998), (16384 + 15386, 499)):
        nxt = 0
        for lo, a, b in long_segments(T, halo):
            assert a == nxt and b > a and 0 <= lo and lo + MAX_PLAN_WINDOW <= T
            assert (lo == 0 or a - lo >= halo) and (lo + MAX_PLAN_WINDOW == T or lo + MAX_PLAN_WINDOW - b >= halo)
            nxt = b
        assert nxt == T
    rng = np.random.default_rng(77)
    with _default_library(lib):
        sfreq, T = 4000, 40000
        t = np.arange(T) / sfreq
        x = np.sin(2 * np.pi * t * np.array([[10.0], [50.0]])) + 0.3 * rng.standard_normal((2, T)) + 40.0
        f_ranges = [[4, 8], [8, 12], [13, 35], [60, 200], [200, 500]]
        for flen, lt, f_ranges in (("999ms", 4, f_ranges), ("1999ms", 8, [[13, 35]]), ("3999ms", 8, [[13, 35]])):
            f = features.MNEFilter(f_ranges, sfreq, filter_length=flen, l_trans_bandwidth=lt, h_trans_bandwidth=lt)
            y = f.filter_data(x)
            assert y.shape == (2, len(f_ranges), T)
            from scipy.signal import fftconvolve   # filter/mne_filter.py:118-124: "same" convolution per filter

            want = np.stack([np.stack([fftconvolve(row, np.asarray(tp, float), "same") for tp in f.filter_bank])
                             for row in x])
            np.testing.assert_allclose(y, want, rtol=0, atol=2e-5 * np.abs(want).max())
            y1 = f.filter_data(x[1])
            assert y1.shape == (1, len(f_ranges), T)
            np.testing.assert_array_equal(y1[0], y[1])
        for fs in (150, 200, 500):
            d = rng.random(fs)
            nf = NotchFilter(fs, 50)
            got = nf.process(d)
            assert got.shape == (fs,)
            want = orc.NotchFilter(fs, 50, taps=nf.filter_bank).process(d)
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * np.abs(want).max())
        xl = rng.standard_normal((3, 40000)) * 20 + np.array([[0.0], [300.0], [-50.0]])
        nf = NotchFilter(1000.0, 50)
        want = orc.NotchFilter(1000.0, 50, taps=nf.filter_bank).process(xl)
        np.testing.assert_allclose(nf.process(xl), want, rtol=0, atol=2e-5 * np.abs(want).max())
        s = NMSettings.get_default()
        s.preprocessing_filter.bandstop_filter = s.preprocessing_filter.lowpass_filter = True
        s.preprocessing_filter.bandpass_filter = False
        s = s.validate()
        pf = PreprocessingFilter(s, 1000.0)
        want = orc.PreprocessingFilter(s, 1000.0, taps=fir_design.preprocessing_filter_bank(s.preprocessing_filter, 1000.0)).process(xl)
        np.testing.assert_allclose(pf.process(xl), want, rtol=0, atol=2e-5 * np.abs(want).max())


def case_plugin_classes_as_the_reference_uses_them(lib):
    """The calls the reference's loop and its tests make on ONE feature class: windows of two lengths from a sampling
    rate that is not a whole number of samples per segment (stream/generator.py:41-53, tests/test_timing.py:43-76: 332 /
    333 samples behind the resampler) with the burst history carried from length to length, an empty channel list
    (tests/test_sharpwave.py:46-62), and pydantic's ValidationError for a burst band that is not defined
    (features/bursts.py:68-73, tests/test_bursts.py:9-13)."""
    import pytest
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings, features

    from tests import parity

    rng = np.random.default_rng(78)
    with _default_library(lib):
        s = NMSettings.get_default()
        s.segment_length_features_ms = 333
        s.fft_settings.windowlength_ms = 300
        s = s.validate()
        ch, sfreq = ["a", "b"], 1000.0
        hj, ohj = features.Hjorth(s, ch, sfreq), orc.Hjorth(s, ch, sfreq)
        ff, off = features.FFT(s, ch, sfreq), orc.FFT(s, ch, sfreq)
        for n in (333, 332, 332, 333, 334, 333):
            x = rng.standard_normal((2, n)) * 10 + np.sin(2 * np.pi * 22 * np.arange(n) / sfreq) * 30
            for got, want in ((hj.calc_feature(x), ohj.calc_feature(x)), (ff.calc_feature(x), off.calc_feature(x))):
                assert list(got) == list(want)
                n_bad, rep, _ = parity.compare(list(got), np.array(list(got.values())), list(want.values()), s, sfreq,
                                               40.0, n, verifier=parity.Verifier(s, ch, sfreq, x))
                assert n_bad == 0, rep
        # burst history and thresholds travel with the stream across the lengths
        sb = NMSettings.get_default()
        sb.segment_length_features_ms = 1000
        sb = sb.validate()
        fs = 1111.111
        bu, obu = features.Bursts(sb, ch, fs), orc.Bursts(sb, ch, fs)
        t0, misses, total = 0, 0, 0
        for hop in range(40):
            n = 1111 + (hop % 3 == 1)
            tt = (t0 + np.arange(n)) / fs
            t0 += 111
            amp = 1.0 + 0.8 * np.sin(2 * np.pi * 0.7 * tt)
            x = np.stack([amp * np.sin(2 * np.pi * 18 * tt), amp * np.sin(2 * np.pi * 70 * tt + 1)]) * 20 \
                + rng.standard_normal((2, n))
            got, want = bu.calc_feature(x), obu.calc_feature(x)
            assert list(got) == list(want)
            g, w = np.array(list(got.values())), np.array(list(want.values()))
            bad = ~np.isclose(g, w, rtol=1e-4, atol=1e-6)
            misses += int(bad.sum())
            total += g.size
        assert len(bu._engines) == 2
        assert misses <= 0.01 * total, (misses, total)   # (a threshold crossing decided in fp32 may differ)
        # an empty channel list: the settings are still checked, nothing is computed
        sw = features.SharpwaveAnalyzer(NMSettings.get_default().validate(), [], 1000.0)
        assert sw.calc_feature(np.zeros((0, 1000))) == {}
        bad = NMSettings.get_default()
        bad.fft_settings.windowlength_ms = 2000
        with pytest.raises(AssertionError):
            features.FFT(bad, [], 1000.0)
        wrong = NMSettings.get_default()
        wrong.bursts_settings.frequency_bands = ["wrong_band"]
        try:
            from pydantic import ValidationError as expected
        except ImportError:
            from py_neuromodulation_amd.settings import SettingsError as expected
        with pytest.raises(expected):
            features.Bursts(wrong, ["ch1", "ch2"], 1000)
        with pytest.raises(ValueError):
            features.Bursts(wrong, ["ch1", "ch2"], 1000)


def case_standalone_rereferencer_float64(lib):
    """The stand-alone ReReferencer is `ref_matrix @ data` in float64 (processing/rereference.py:88-102): the reference's
    tests hold it to rtol 1e-7 against float64 arithmetic (tests/test_rereference.py:57-182); nmx_reref_f64 keeps float64
    from the caller's array to the result, for any number of samples, strided rows, IEEE NaN / inf semantics."""
    import pandas as pd
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.processing import ReReferencer

    rng = np.random.default_rng(91)
    n = 21
    names = [f"c{i}" for i in range(n)]
    types = ["ecog"] * 9 + ["dbs"] * 8 + ["seeg"] * 4
    ref = ["average"] * 9 + [f"c{9 + (i + 1) % 8}" for i in range(8)] + ["None", "c17&c19", "average", "average"]
    ch = pd.DataFrame({"name": names, "rereference": ref, "used": [1] * n, "target": [0] * n, "type": types,
                       "status": ["good"] * 20 + ["bad"], "new_name": names})
    with _default_library(lib):
        rr = ReReferencer(1000.0, ch)
        R = chmod.reref_matrix(chmod.load_channels(ch))
        n = R.shape[1]   # (the used rows of the table)
        assert R.shape == (n, n) and n >= 20
        for T in (1, 255, 1000, 40001):
            x = rng.standard_normal((n, T)) * 50 + rng.uniform(-4000, 4000, (n, 1))
            got = rr.process(x)
            assert got.dtype == np.float64 and got.shape == (n, T)
            np.testing.assert_allclose(got, R @ x, rtol=1e-12, atol=1e-11)
        big = rng.standard_normal((2 * n, 3000)) * 1e3
        view = big[::2, 500:2500]                      # strided rows, offset start
        np.testing.assert_allclose(rr.process(view), R @ view, rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(rr.process(np.asfortranarray(x[:, :300])), R @ x[:, :300], rtol=1e-12, atol=1e-11)
        x = rng.standard_normal((n, 64))
        x[3, 5], x[12, 7], x[0, 9] = np.nan, np.inf, -np.inf
        with np.errstate(invalid="ignore"):
            want = R @ x
        got = rr.process(x)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-12, atol=1e-11)
        import pytest

        with pytest.raises(ValueError):
            rr.process(x[:5])
        assert rr.process(np.zeros((n, 0))).shape == (n, 0)


def case_standalone_resampler_float64(lib):
    """The stand-alone Resampler is mne.filter.resample on the float64 array (processing/resample.py:42-60), any length:
    the reference's tests resample 10 s at 4 kHz and at 1 kHz in one call (tests/test_nm_resample.py:8-47).
    nmx_resample_f64 against the oracle's restatement: power-of-two and Bluestein lengths, odd / prime padded lengths,
    windows of 1 - 5 samples, N-D input, NaN."""
    from oracle import mne_restated as mr
    from py_neuromodulation_amd.processing import Resampler

    rng = np.random.default_rng(92)
    with _default_library(lib):
        for fs, to, T in ((4000.0, 1000.0, 40000), (1000.0, 4000.0, 10000), (1375.0, 1000.0, 1375), (1000.0, 900.0, 333),
                          (2048.0, 1000.0, 5000), (1000.0, 1111.0, 999), (1000.0, 500.0, 5), (1000.0, 3000.0, 1),
                          (1000.0, 250.0, 2), (22050.0, 1000.0, 22050), (1000.0, 999.0, 70001)):
            x = rng.standard_normal((3, T)) * 10 + rng.uniform(-1e3, 1e3, (3, 1))
            got = Resampler(fs, to).process(x)
            want = mr.resample(x, up=to / fs, down=1.0)
            assert got.shape == want.shape and got.dtype == np.float64
            if want.size:
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max())
        r = Resampler(4000.0, 1000.0)
        t = np.linspace(0, 10, 40000)
        data = np.sin(2 * np.pi * t * np.arange(10, 51, 10)[:, None])        # (tests/test_nm_resample.py:8-30)
        out = r.process(data)
        assert out.shape == (5, 10000)
        np.testing.assert_allclose(out[:, 50:-50], data[:, ::4][:, 50:-50], atol=2e-3)
        one = r.process(data[2])                                            # 1-D in, 1-D out
        assert one.shape == (10000,)
        np.testing.assert_array_equal(one, out[2])
        cube = rng.standard_normal((2, 3, 400))
        np.testing.assert_allclose(r.process(cube), mr.resample(cube.reshape(6, 400), up=0.25).reshape(2, 3, 100),
                                   rtol=0, atol=1e-12)
        bad = data[:2, :4000].copy()
        bad[1, 77] = np.nan
        y = r.process(bad)
        assert np.isnan(y[1]).all() and not np.isnan(y[0]).any()
        same = Resampler(1000.0, 1000.0)
        assert same.process(data) is data


def case_processor_hop_by_hop_with_two_window_lengths(lib):
    """`DataProcessor.process(window)` under a loop that hands it whatever the generator cuts (the reference's own
    `Stream.run`, stream/stream.py:280-311, at 1111.111 Hz: windows of 1111 and 1112 samples) == `Stream.run` of the same
    recording (one plan per length, hops in order): the burst history and Kalman filters travel between the plans at every
    change of length, one feature normaliser serves them all.  Without the normaliser bit for bit; with the default
    z-score to the composition tolerance (scan over a batch vs one hop at a time).  Plus the methods the reference's
    Stream calls on its processor after the loop (stream/data_processor.py:313-351)."""
    import json
    import tempfile
    from pathlib import Path

    from py_neuromodulation_amd import DataProcessor, NMSettings, Stream
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.generator import window_schedule

    sfreq = 1111.111
    rng = np.random.default_rng(314)
    T = int(7.5 * sfreq)
    t = np.arange(T) / sfreq
    amp = 1.0 + 0.7 * np.sin(2 * np.pi * 0.4 * t)
    data = np.stack([amp * np.sin(2 * np.pi * 17 * t), amp * np.sin(2 * np.pi * 25 * t + 1), np.sin(2 * np.pi * 70 * t)]) * 30 \
        + rng.standard_normal((3, T)) * 4 + np.array([[0.0], [900.0], [-40.0]])
    ch = chmod.get_default_channels_from_data(data)
    for norm in (False, True):
        s = NMSettings.get_fast_compute()
        s.features.bursts = s.features.raw_hjorth = s.features.bandpass_filter = True
        s.bandpass_filter_settings.kalman_filter = True
        s.postprocessing.feature_normalization = norm
        s = s.validate()
        starts, lens, _ = window_schedule(T, sfreq, s.sampling_rate_features_hz, s.segment_length_features_ms)
        assert len(set(lens.tolist())) == 2 and len(starts) > 40
        df = Stream(sfreq=sfreq, channels=ch, settings=s, lib=lib, line_noise=50).run(data, save_csv=False)
        dp = DataProcessor(sfreq=sfreq, settings=s, channels=ch, line_noise=50, lib=lib, verbose=False)
        rows = [dp.process(data[:, a:a + n]) for a, n in zip(starts, lens)]
        assert list(rows[0]) == list(df.columns[:-1]) and len(dp._by_len) == 2
        got = np.array([list(r.values()) for r in rows])
        want = df.to_numpy(dtype=np.float64)[:, :-1]
        if not norm:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)
        dp.reset()                                                     # a second pass from a fresh state: the same rows
        again = np.array([list(dp.process(data[:, a:a + n]).values()) for a, n in zip(starts[:12], lens[:12])])
        np.testing.assert_array_equal(again, got[:12])
    with tempfile.TemporaryDirectory() as tmp:
        dp.save_sidecar(tmp, "sub", {"sess_right": None})
        dp.save_settings(tmp, "sub")
        dp.save_channels(tmp, "sub")
        dp.save_features(df, tmp, "sub")
        side = json.loads((Path(tmp) / "sub" / "sub_SIDECAR.json").read_text())
        assert side == {"original_fs": sfreq, "final_fs": sfreq // 1, "sfreq": s.sampling_rate_features_hz, "sess_right": None}
        assert sorted(p.name for p in (Path(tmp) / "sub").iterdir()) == ["sub_SETTINGS.yaml", "sub_SIDECAR.json", "sub_channels.csv"]
        assert (Path(tmp) / "sub_FEATURES.csv").exists()

"""CPU-only check of the KERNEL LOGIC: the HIP kernel source compiled single-threaded with g++
(tests/emu/nmx_emu.cpp, test-only) behind the same C ABI, driven through the same Python host
code, compared with the reference-generated goldens at fp32 tolerances (tests/parity.py).

What this does NOT cover (only the -m gpu tests do): wave/barrier behaviour, LDS races,
shuffles, occupancy -- and it is not a product code path: the package never loads this library.
"""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from tests import parity  # noqa: E402


@pytest.fixture(scope="module")
def emu_lib():
    import __graft_entry__ as ge
    from py_neuromodulation_amd import _lib

    return _lib.NmxLibrary(ge.build_emu())


from tests import parity_cases as pc  # noqa: E402


@pytest.mark.parametrize("case", pc.FEATURE_CASES)
def test_feature_cases_match_reference_goldens(emu_lib, case):
    pc.case_feature_cases_match_reference_goldens(emu_lib, case)


def test_dc_offsets_1e3_and_1e5_at_stated_tolerances(emu_lib):
    pc.case_dc_offsets(emu_lib)


def test_pipelined_f64_batch_equals_plain_batch(emu_lib):
    pc.case_pipelined_f64_batch(emu_lib)


def test_multi_device_pipelined_batch_equals_passes_in_a_row(emu_lib):
    pc.case_multi_device_pipelined_batch(emu_lib)


def test_sharpwave_reference_test_inputs(emu_lib):
    pc.case_sharpwave_reference_test_inputs(emu_lib)


def test_bursts_sequence_state_across_batches(emu_lib):
    pc.case_bursts_sequence_state_across_batches(emu_lib)


def test_preprocessing_notch_and_reref(emu_lib):
    pc.case_preprocessing_notch_and_reref(emu_lib)


def test_filter_window_matches_mnefilter_shape_and_values(emu_lib):
    pc.case_filter_window_matches_mnefilter_shape_and_values(emu_lib)


def test_nan_mask_and_clean_on_load(emu_lib):
    pc.case_nan_mask_and_clean_on_load(emu_lib)


def test_pipeline_readme_no_normalisation(emu_lib):
    pc.case_pipeline_readme_no_normalisation(emu_lib)


def test_pipeline_readme_default_zscore(emu_lib):
    pc.case_pipeline_readme_default_zscore(emu_lib)


def test_pipeline_nan_and_channel_table(emu_lib):
    pc.case_pipeline_nan_and_channel_table(emu_lib)


def test_bursts_steady_state_vs_oracle(emu_lib):
    pc.case_bursts_steady_state_vs_oracle(emu_lib)


def test_ragged_float_sfreq_stream(emu_lib):
    pc.case_ragged_float_sfreq_stream(emu_lib)


def test_odd_windows_and_spectra(emu_lib):
    pc.case_odd_windows_and_spectra(emu_lib)


def test_feature_normalizer_batches(emu_lib):
    pc.case_feature_normalizer_batches(emu_lib)


def test_feature_normalizer_power(emu_lib):
    pc.case_feature_normalizer_power(emu_lib)


def test_raw_quantile_subsample(emu_lib):
    pc.case_raw_quantile_subsample(emu_lib)


def test_resampler_long_windows(emu_lib):
    pc.case_resampler_long_windows(emu_lib)


def test_high_rate_partitioned_fir(emu_lib):
    pc.case_high_rate_partitioned_fir(emu_lib)


@pytest.mark.parametrize("seed", list(range(3)) + [102790])
def test_random_settings_highrate(emu_lib, seed):
    pc.case_random_settings_highrate(emu_lib, seed)


def test_config5_degenerate_bursts_and_welch(emu_lib):
    pc.case_config5_degenerate(emu_lib)


def test_stream_output_files(emu_lib, tmp_path):
    pc.case_stream_output_files(emu_lib, tmp_path)


def test_bandpower_kalman_sequence(emu_lib):
    pc.case_bandpower_kalman_sequence(emu_lib)


def test_resampler(emu_lib):
    pc.case_resampler(emu_lib)


def test_preprocessing_filter(emu_lib):
    pc.case_preprocessing_filter(emu_lib)


def test_config5_30khz_512pt(emu_lib):
    pc.case_config5_30khz_512pt(emu_lib)


def test_raw_normalizer(emu_lib):
    pc.case_raw_normalizer(emu_lib)


def test_raw_normalizer_order_methods(emu_lib):
    pc.case_raw_normalizer_order_methods(emu_lib)


def test_psd_keys_skip_normalisation(emu_lib):
    pc.case_psd_keys_skip_normalisation(emu_lib)


def test_reref_structured_matrices(emu_lib):
    pc.case_reref_structured_matrices(emu_lib)


def test_raw_resampling_reference_quirk(emu_lib):
    pc.case_raw_resampling_reference_quirk(emu_lib)


def test_windows_shorter_than_the_spectral_segment(emu_lib):
    pc.case_short_windows(emu_lib)


@pytest.mark.parametrize("seed", pc.RANDOM_SETTINGS_SEEDS[:12])
def test_random_settings_stream_equals_oracle(emu_lib, seed):
    pc.case_random_settings(emu_lib, seed)


@pytest.mark.parametrize("seed", pc.WIDE_SETTINGS_SEEDS[:12])
def test_random_settings_wide_stream_equals_oracle(emu_lib, seed):
    pc.case_random_settings_wide(emu_lib, seed)


@pytest.mark.parametrize("seed", pc.CHANNEL_TABLE_SEEDS[:8])
def test_random_channel_tables_stream_equals_oracle(emu_lib, seed):
    pc.case_random_channel_tables(emu_lib, seed)


@pytest.mark.parametrize("seed", pc.BURST_STREAM_SEEDS[:6])
def test_random_burst_streams_equal_oracle(emu_lib, seed):
    pc.case_random_burst_streams(emu_lib, seed)


@pytest.mark.parametrize("seed", pc.RANDOM_SETTINGS_SEEDS[:8])
def test_random_settings_window_by_window_equals_batch(emu_lib, seed):
    pc.case_random_window_by_window(emu_lib, seed)


def test_burst_fill_walk_equals_workgroup_walk(emu_lib, monkeypatch):
    """The sort-once walk of a fresh stream's fill phase (nmx_k_burst_fill.h: slot look-up with the claim mask, arrival
    mask, moving rank pointer -- here its single-thread form) against the per-hop merge walk: identical burst features
    over a stream that crosses the point where the 3 s history is full, in one batch and in uneven batches, with a
    quantised channel (runs of equal values)."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.bursts_settings.time_duration_s = 3
    s = s.validate()
    C, W, hop, n = 3, 1000, 100, 80
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((C, W + (n - 1) * hop)) * 20).astype(np.float32)
    x[2] = np.round(x[2])
    ch = [f"c{i}" for i in range(C)]
    starts = np.arange(n) * hop

    def run(fill, plan):
        monkeypatch.setenv("NMX_THR_FILL", "1" if fill else "0")
        eng = HotPathEngine(s, ch, 1000.0, lib=emu_lib, features=["bursts"], bank_taps=None)
        out, i = [], 0
        for k in plan:
            out.append(eng.process_batch(x, starts[i:i + k]))
            i += k
        eng.close()
        return np.concatenate(out)

    want = run(False, [n])
    assert not np.isnan(want).any()
    for plan in ([n], [5, 1, n - 6], [30, 50]):
        np.testing.assert_array_equal(run(True, plan), want)


def test_user_registered_features(emu_lib):
    pc.case_user_features(emu_lib)


def test_ragged_windows_carry_burst_and_kalman_state(emu_lib):
    pc.case_ragged_bursts(emu_lib)


def test_long_windows_sharp_waves_and_order_normalisers(emu_lib):
    pc.case_long_windows(emu_lib)


def test_raw_normalisation_with_ragged_window_lengths(emu_lib):
    pc.case_ragged_rawnorm(emu_lib)


def test_raw_order_normalisers_with_lists_in_device_memory(emu_lib, monkeypatch):
    monkeypatch.setenv("NMX_RAWNORM_GLOBAL_LISTS", "1")
    pc.case_raw_normalizer_order_methods(emu_lib)


@pytest.mark.parametrize("large", [False, True])
def test_input_layouts_give_identical_results(emu_lib, large):
    pc.case_input_layouts(emu_lib, large=large)


def test_input_layouts_single_channel(emu_lib):
    pc.case_input_layouts_single_channel(emu_lib)


@pytest.mark.parametrize("devices", [None, (0, 0)])
def test_real_recording_of_the_reference_tests(emu_lib, devices):
    pc.case_real_recording(emu_lib, devices=devices)


@pytest.mark.parametrize("tag", ["", "stft_"])
def test_reref_group_members_on_the_rail(emu_lib, tag):
    pc.case_inf_members(emu_lib, tag=tag, spectral_nan_ok=False)


def test_failed_run_leaves_no_side_files(emu_lib, tmp_path, monkeypatch):
    """The reference writes the side-car / settings / channels files after its loop (stream/stream.py:338); the batch driver
    writes them on a thread next to the device work -- a run that raises must not leave them behind."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.data_processor import DataProcessor
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.raw_hjorth = True
    data = np.random.default_rng(3).standard_normal((3, 2500))
    st = Stream(sfreq=1000.0, data=data, settings=s, lib=emu_lib)

    def boom(self, *a, **k):
        raise RuntimeError("device lost")

    monkeypatch.setattr(DataProcessor, "process_batch", boom)
    with pytest.raises(RuntimeError, match="device lost"):
        st.run(out_dir=tmp_path, experiment_name="subx", save_csv=False)
    assert not any((tmp_path / "subx").glob("*")) if (tmp_path / "subx").exists() else True
    monkeypatch.undo()
    st.run(out_dir=tmp_path, experiment_name="subx", save_csv=False)
    assert sorted(p.name for p in (tmp_path / "subx").iterdir()) == ["subx_SETTINGS.yaml", "subx_SIDECAR.json", "subx_channels.csv"]


def test_trends_are_counted_not_hidden(emu_lib):
    acc = pc.case_trends(emu_lib)
    assert sum(acc.values()) <= 40, acc   # (emulator 11, all Welch bins at 1e-4 of the swell's leakage; see the budget file)


def test_matrix_pipe_spectrum_kernel_arithmetic_and_flags(emu_lib):
    """The emulator body of nmx_k_specmm.h (twice-folded half-sample-phase contraction against the HOST's table, bin
    bookkeeping, single-pass time domain, the overflow flag and the redo of flagged windows): config[1]'s feature set on
    windows with offsets, a NaN stretch, an infinity, a flat channel; odd channel count, a ragged last tile -- against the
    float64 oracle, and the plan must say which path it took."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = s.features.return_raw = True
    C, n = 5, 19
    rng = np.random.default_rng(3)
    T = 1000 + (n - 1) * 100
    x = (rng.standard_normal((C, T)) * 50 + rng.uniform(-500, 500, (C, 1))).astype(np.float32)
    x[1] = 0.0
    x[2, 1500:1507] = np.nan
    x[3, 700] = np.inf
    ch = [f"ch{i}" for i in range(C)]
    eng = HotPathEngine(s, ch, 1000.0, lib=emu_lib)
    assert bool(eng.desc.features) and eng.process_batch(x, np.arange(n) * 100).shape == (n, eng.n_outputs)
    got = eng.process_batch(x, np.arange(n) * 100)
    one = np.stack([eng.process_window(x[:, a:a + 1000].astype(np.float64)) for a in (0, 900, 1800)])
    np.testing.assert_array_equal(got[[0, 9, 18]], one)
    feats = [orc._FEATURE_CLS[f](s, ch, 1000.0) for f in s.features.get_enabled()]
    for i in range(n):
        w = np.nan_to_num(x[:, i * 100:i * 100 + 1000].astype(np.float64))
        want = {}
        for f in feats:
            want.update(f.calc_feature(w))
        assert list(want) == list(eng.keys)
        keep = [k for k, key in enumerate(eng.keys) if not (key.startswith("ch3_") and i * 100 <= 700)]   # (ch3 on the rail there)
        wv = list(want.values())
        n_bad, rep, _ = parity.compare([eng.keys[k] for k in keep], got[i][keep], [wv[k] for k in keep], s, 1000.0, 600.0, 1000,
                                       verifier=parity.Verifier(s, ch, 1000.0, w))
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()


def test_nan_on_an_offset_channel_without_a_rereference(emu_lib):
    pc.case_dc_nan(emu_lib)


def test_standalone_classes_any_length(emu_lib):
    pc.case_standalone_classes_any_length(emu_lib)


def test_plugin_classes_as_the_reference_uses_them(emu_lib):
    pc.case_plugin_classes_as_the_reference_uses_them(emu_lib)


def test_standalone_rereferencer_float64(emu_lib):
    pc.case_standalone_rereferencer_float64(emu_lib)


def test_standalone_resampler_float64(emu_lib):
    pc.case_standalone_resampler_float64(emu_lib)


def test_processor_hop_by_hop_with_two_window_lengths(emu_lib):
    pc.case_processor_hop_by_hop_with_two_window_lengths(emu_lib)

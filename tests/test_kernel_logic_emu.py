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

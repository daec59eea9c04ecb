"""GPU parity tests proper (-m gpu): the HIP kernels behind the C ABI (libnmx.so, loaded by the
package's own loader -- no emulator, no fallback) against the reference-generated goldens and,
at sizes beyond the goldens, against the CPU oracle on the same seeded inputs.
Tolerances: tests/parity.py."""

import numpy as np
import pytest

from tests import parity
from tests import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_lib():
    from py_neuromodulation_amd import _lib

    lib = _lib.get_library()
    assert lib.device_count() >= 1, "no HIP device visible"
    assert "libnmx.so" in str(lib.path)
    return lib


@pytest.mark.parametrize("case", pc.FEATURE_CASES)
def test_feature_cases_match_reference_goldens(gpu_lib, case):
    pc.case_feature_cases_match_reference_goldens(gpu_lib, case)


def test_sharpwave_reference_test_inputs(gpu_lib):
    pc.case_sharpwave_reference_test_inputs(gpu_lib)


def test_bursts_sequence_state_across_batches(gpu_lib):
    pc.case_bursts_sequence_state_across_batches(gpu_lib)


def test_preprocessing_notch_and_reref(gpu_lib):
    pc.case_preprocessing_notch_and_reref(gpu_lib)


def test_filter_window_matches_mnefilter_shape_and_values(gpu_lib):
    pc.case_filter_window_matches_mnefilter_shape_and_values(gpu_lib)


def test_nan_mask_and_clean_on_load(gpu_lib):
    pc.case_nan_mask_and_clean_on_load(gpu_lib)


def test_batch_equals_window_by_window_and_oracle_64ch(gpu_lib):
    """BASELINE config[1]: 64 ch @ 1 kHz, W=1000, hop=100, FFT + Hjorth + LineLength; the
    batch path must equal the one-window path bit for bit and the oracle within tolerance."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = True
    sfreq, C, T = 1000.0, 64, 4000
    rng = np.random.default_rng(1234)
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t)
         + 5 * np.sin(2 * np.pi * 70 * t) + rng.uniform(-500, 500, (C, 1))).astype(np.float32)
    ch = [f"ch{i}" for i in range(C)]
    starts, _, _ = orc.window_schedule(T, sfreq, 10, 1000)
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    got = eng.process_batch(x, starts)
    one = np.stack([eng.process_window(x[:, a:a + 1000].astype(np.float64)) for a in starts])
    np.testing.assert_array_equal(got, one)
    feats = [orc.Hjorth(s, ch, sfreq), orc.FFT(s, ch, sfreq), orc.LineLength(s, ch, sfreq)]
    for i, a in enumerate(starts):
        want = {}
        for f in feats:
            want.update(f.calc_feature(x[:, a:a + 1000].astype(np.float64)))
        assert list(want) == eng.keys
        n_bad, rep, _ = parity.compare(eng.keys, got[i], list(want.values()), s, sfreq, 300.0, 1000)
        assert n_bad == 0, rep
    eng.close()


def test_pipeline_readme_no_normalisation(gpu_lib):
    pc.case_pipeline_readme_no_normalisation(gpu_lib)


def test_pipeline_readme_default_zscore(gpu_lib):
    pc.case_pipeline_readme_default_zscore(gpu_lib)


def test_pipeline_nan_and_channel_table(gpu_lib):
    pc.case_pipeline_nan_and_channel_table(gpu_lib)


def test_bursts_steady_state_vs_oracle(gpu_lib):
    pc.case_bursts_steady_state_vs_oracle(gpu_lib)

"""GPU parity tests proper (-m gpu): the HIP kernels behind the C ABI (libnmx.so, loaded by the
package's own loader -- no emulator, no fallback) against the reference-generated goldens and,
at sizes beyond the goldens, against the CPU oracle on the same seeded inputs.
Tolerances: tests/parity.py."""

import os

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


def test_dc_offsets_1e3_and_1e5_at_stated_tolerances(gpu_lib):
    pc.case_dc_offsets(gpu_lib)


def test_pipelined_f64_batch_equals_plain_batch(gpu_lib):
    pc.case_pipelined_f64_batch(gpu_lib)


def test_multi_device_pipelined_batch_equals_passes_in_a_row(gpu_lib):
    pc.case_multi_device_pipelined_batch(gpu_lib)


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
        ver = parity.Verifier(s, ch, sfreq, x[:, a:a + 1000].astype(np.float64))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], list(want.values()), s, sfreq, 300.0, 1000, verifier=ver)
        assert n_bad == 0, rep
    eng.close()


def test_matrix_pipe_spectrum_kernel(gpu_lib, monkeypatch):
    """nmx_kern_specmm_w1000 (nmx_k_specmm.h): the FFT band means of BASELINE config[1] as a pruned DFT on the matrix
    pipe, Hjorth / LineLength / Raw as single-pass per-lane statistics next to it.  70 channels x 45 hops (a ragged last
    tile of 32-window groups), per-channel offsets of +-500, a NaN stretch (nan_to_num on load) and a flat channel;
    against the float64 oracle under the standard policy, and against the wave-level FFT kernel it replaces."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = s.features.return_raw = True
    sfreq, C, n_hops = 1000.0, 70, 45
    T = 1000 + (n_hops - 1) * 100
    rng = np.random.default_rng(99)
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + rng.uniform(-500, 500, (C, 1))).astype(np.float32)
    x[5] = 0.0
    x[9, 2000:2010] = np.nan
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * 100
    monkeypatch.setenv("NMX_SPECMM", "1")   # (opt-in: see nmx_specmm.hip)
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    got = eng.process_batch(x, starts)
    assert "nmx_kern_specmm_w1000" in eng.kernels(2), eng.kernels(2)
    one = np.stack([eng.process_window(x[:, a:a + 1000].astype(np.float64)) for a in starts[:3]])
    np.testing.assert_array_equal(got[:3], one)     # one window == the batch, bit for bit
    # window starts of every residue mod 4 (an odd hop): the same kernel, the same values as window by window
    odd = np.arange(1, T - 1000, 101)
    got_odd = eng.process_batch(x, odd)
    assert "nmx_kern_specmm_w1000" in eng.kernels(2), eng.kernels(2)
    one_odd = np.stack([eng.process_window(x[:, a:a + 1000].astype(np.float64)) for a in odd[:4]])
    np.testing.assert_array_equal(got_odd[:4], one_odd)
    eng.close()
    monkeypatch.setenv("NMX_SPECMM", "0")
    eng0 = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    old = eng0.process_batch(x, starts)
    assert "specmm" not in eng0.kernels(2)
    eng0.close()
    feats = [orc._FEATURE_CLS[f](s, ch, sfreq) for f in s.features.get_enabled()]
    for i in (0, 7, 19, 20, 21, 44):
        w = np.nan_to_num(x[:, starts[i]:starts[i] + 1000].astype(np.float64))
        want = {}
        for f in feats:
            want.update(f.calc_feature(w))
        assert list(want) == eng.keys
        ver = parity.Verifier(s, ch, sfreq, w)
        for tag, rows in (("specmm", got), ("wave fft", old)):
            n_bad, rep, _ = parity.compare(eng.keys, rows[i], list(want.values()), s, sfreq, 600.0, 1000, verifier=ver)
            assert n_bad == 0, f"{tag} hop {i}\n{rep}"


@pytest.mark.parametrize("C,n_hops", [(1, 1), (1, 17), (3, 15), (3, 16), (3, 4095), (255, 17), (257, 1), (257, 16)])
def test_matrix_pipe_spectrum_kernel_ragged_shapes(gpu_lib, C, n_hops):
    """nmx_kern_specmm_w1000 works on tiles of 16 consecutive windows of one channel, four tiles per workgroup, one
    persistent workgroup per CU: channel counts around a multiple of the wave count and window counts around a multiple
    of the tile (1, 15, 16, 17, 4095: a last tile with 1 / 15 live columns), with a NaN and an infinity inside the LAST
    tile (its `todo` mask: the wave-level kernel redoes exactly those windows) and in the first.  Every window against
    the float64 oracle under the standard policy."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.fft = s.features.raw_hjorth = s.features.linelength = s.features.return_raw = True
    sfreq, hop = 1000.0, 8 if n_hops > 1000 else 100
    T = 1000 + (n_hops - 1) * hop
    rng = np.random.default_rng(1000 * C + n_hops)
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + rng.uniform(-500, 500, (C, 1))).astype(np.float32)
    x[C - 1, T - 3] = np.nan            # the last window(s) of the last channel
    x[0, 5] = np.inf                    # the first window of the first
    if n_hops > 16:
        x[C // 2, 16 * hop + 999] = -np.inf   # a window of the second tile only
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * hop
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    got = eng.process_batch(x, starts)
    assert "nmx_kern_specmm_w1000" in eng.kernels(2), eng.kernels(2)
    keys = list(eng.keys)
    eng.close()
    feats = [orc._FEATURE_CLS[f](s, ch, sfreq) for f in s.features.get_enabled()]
    # (4095 hops: every 13th window and everything around the tile / batch ends)
    pick = range(n_hops) if n_hops <= 64 else sorted(set(range(0, n_hops, 13)) | set(range(40)) | set(range(n_hops - 40, n_hops)))
    for i in pick:
        w = np.nan_to_num(x[:, starts[i]:starts[i] + 1000].astype(np.float64))
        want = {}
        for f in feats:
            want.update(f.calc_feature(w))
        assert list(want) == keys
        # a channel with an infinity in this window sits on the rail (+-FLT_MAX here, +-DBL_MAX there): its features are
        # rail-derived on both scales (tests/parity_cases.py: case_inf_members) -- here only "not NaN, not ordinary"
        railed = {ch[c] for c in np.flatnonzero(np.isinf(x[:, starts[i]:starts[i] + 1000]).any(axis=1))}
        keep = [k for k, key in enumerate(keys) if not any(key.startswith(r + "_") for r in railed)]
        for k, key in enumerate(keys):   # (a transform of a window with a sample on the rail: huge or +inf)
            if k not in keep and ("Activity" in key or "LineLength" in key or "_fft_" in key):
                assert abs(float(got[i][k])) > 1e6, (i, key, got[i][k])   # (huge or +inf; never NaN: NmxBandAcc::railed)
        assert not np.isnan(got[i][keep]).any()
        ver = parity.Verifier(s, ch, sfreq, w)
        wv = list(want.values())
        n_bad, rep, _ = parity.compare([keys[k] for k in keep], got[i][keep], [wv[k] for k in keep], s, sfreq, 600.0, 1000,
                                       verifier=ver)
        assert n_bad == 0, f"C = {C}, {n_hops} hops, hop {i}\n{rep}"


def test_pipeline_readme_no_normalisation(gpu_lib):
    pc.case_pipeline_readme_no_normalisation(gpu_lib)


def test_pipeline_readme_default_zscore(gpu_lib):
    pc.case_pipeline_readme_default_zscore(gpu_lib)


def test_pipeline_nan_and_channel_table(gpu_lib):
    pc.case_pipeline_nan_and_channel_table(gpu_lib)


def test_bursts_steady_state_vs_oracle(gpu_lib):
    pc.case_bursts_steady_state_vs_oracle(gpu_lib)


def test_bursts_steady_state_list_in_l2(gpu_lib, monkeypatch):
    """The one-wave threshold walk with its top-K list in the L2-resident state array (what a stream of more than 4096 hops
    runs; a young stream keeps the list in LDS for the launch): the same case, the same thresholds."""
    monkeypatch.setenv("NMX_THR_LIST_LDS", "0")
    pc.case_bursts_steady_state_vs_oracle(gpu_lib)


def _bench_like_engine(gpu_lib, C, scale=1.0, pre=True, device=0):
    from py_neuromodulation_amd import NMSettings, fir_design
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.bandpass_filter = True
    s.features.stft = True
    ch = [f"ch{i}_avgref" for i in range(C)]
    R = np.full((C, C), -1.0 / (C - 1))
    np.fill_diagonal(R, 1.0)
    eng = HotPathEngine(s, ch, 1000.0, lib=gpu_lib, ref_matrix=R if pre else None,
                        notch_taps=fir_design.notch_bank(1000.0, 50) if pre else None)
    return s, eng


def test_full_size_properties_256ch(gpu_lib):
    """BASELINE headline size (256 ch @ 1 kHz, W=1000, hop=100, all features, notch + CAR), where the
    oracle is too slow to run everything: size-independent properties instead.
      (1) the batch path equals the one-window path bit for bit (windows are strided views);
      (2) scaling the input by 4 scales amplitude-like features by 4, power-like by 16 (log: +log10),
          leaves shape-like features (mobility, complexity, intervals) unchanged;
      (3) three hops are checked against the CPU oracle on 256 channels."""
    from oracle import nm_oracle as orc

    C, n_hops = 256, 48
    T = 1000 + (n_hops - 1) * 100
    rng = np.random.default_rng(1234)
    t = np.arange(T) / 1000.0
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + 5 * np.sin(2 * np.pi * 70 * t)
         + rng.uniform(-500, 500, (C, 1))).astype(np.float32)
    starts = np.arange(n_hops) * 100
    s, eng = _bench_like_engine(gpu_lib, C)
    got = eng.process_batch(x, starts)
    assert not np.isnan(got).any()
    # (1) stateless columns must agree exactly between batch and single-window calls
    s1, eng1 = _bench_like_engine(gpu_lib, C)
    stateless = np.array(["_bursts_" not in k for k in eng.keys])
    for i in (0, 17, 47):
        one = eng1.process_window(x[:, starts[i]:starts[i] + 1000].astype(np.float64))
        np.testing.assert_array_equal(one[stateless], got[i][stateless])
    # (2) scaling
    s4, eng4 = _bench_like_engine(gpu_lib, C)
    got4 = eng4.process_batch(x * 4.0, starts)
    keys = np.array(eng.keys)

    def sel(sub):
        return np.array([sub in k for k in keys])

    np.testing.assert_allclose(got4[:, sel("_raw")], 4 * got[:, sel("_raw")], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(got4[:, sel("_LineLength")], 4 * got[:, sel("_LineLength")], rtol=1e-5)
    np.testing.assert_allclose(got4[:, sel("RawHjorth_Activity")], 16 * got[:, sel("RawHjorth_Activity")], rtol=1e-5)
    np.testing.assert_allclose(got4[:, sel("RawHjorth_Mobility")], got[:, sel("RawHjorth_Mobility")], rtol=1e-5)
    lg = np.log10(4.0)
    for fam in ("_fft_", "_welch_", "_stft_"):
        add = 2 * lg if fam == "_welch_" else lg
        np.testing.assert_allclose(got4[:, sel(fam)], got[:, sel(fam)] + add, rtol=0, atol=2e-4)
    np.testing.assert_allclose(got4[:, sel("_bandpass_activity_")], got[:, sel("_bandpass_activity_")] + 2 * lg,
                               rtol=0, atol=2e-4)
    np.testing.assert_allclose(got4[:, sel("_interval_")], got[:, sel("_interval_")], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(got4[:, sel("_prominence_")], 4 * got[:, sel("_prominence_")], rtol=1e-4, atol=1e-3)
    # (3) oracle on three hops (stateless features; bursts need the whole history)
    names = [f"ch{i}" for i in range(C)]
    channels = {"name": names, "rereference": ["average"] * C, "used": [1] * C, "target": [0] * C,
                "type": ["ecog"] * C, "status": ["good"] * C, "new_name": [f"{n}_avgref" for n in names]}
    s.postprocessing.feature_normalization = False
    s.preprocessing = ["notch_filter", "re_referencing"]
    s.features.bursts = False
    dp = orc.DataProcessor(1000.0, s, channels, line_noise=50)
    pv = parity.PipelineVerifiers(s, channels, 1000.0, x, starts, 1000, line_noise=50)
    okeys = None
    for i in (0, 31):
        want = dp.process(x[:, starts[i]:starts[i] + 1000].astype(np.float64))
        okeys = list(want.keys())
        idx = [eng.keys.index(k) for k in okeys]
        n_bad, rep, _ = parity.compare(okeys, got[i][idx], list(want.values()), s, 1000.0, 300.0, 1000,
                                       verifier=pv.row(i))
        assert n_bad == 0, rep
    for e in (eng, eng1, eng4):
        e.close()


def test_headline_bursts_through_the_ring_256ch(gpu_lib):
    """The bursts columns at the headline width (256 ch @ 1 kHz, all features, notch + CAR over all 256 rows) against
    the oracle THROUGH the 30 s percentile history: 330 hops (the ring is full from hop 291), two batches, 8 of the
    256 channels.  The oracle gets those 8 rows of the common-average-referenced recording (float64 product) and
    runs notch + Bursts hop by hop; every bursts column of those channels is compared under the standard policy
    (decision-margin verifier for env >= thr)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings

    C, n_hops = 256, 330
    T = 1000 + (n_hops - 1) * 100
    rng = np.random.default_rng(77)
    t = np.arange(T) / 1000.0
    amp = 1 + 0.7 * np.sin(2 * np.pi * 0.31 * t)
    x = (rng.standard_normal((C, T)) * 50 + 25 * amp * np.sin(2 * np.pi * 18 * t) + 5 * np.sin(2 * np.pi * 70 * t)
         + rng.uniform(-500, 500, (C, 1))).astype(np.float32)
    starts = np.arange(n_hops) * 100
    s, eng = _bench_like_engine(gpu_lib, C)
    got = np.concatenate([eng.process_batch(x, starts[:200]), eng.process_batch(x, starts[200:])])
    eng.close()
    pick = [0, 3, 64, 65, 127, 128, 200, 255]
    R = np.full((C, C), -1.0 / (C - 1))
    np.fill_diagonal(R, 1.0)
    xr = R[pick] @ x.astype(np.float64)
    names = [f"ch{i}_avgref" for i in pick]
    so = NMSettings.get_default()
    so.features.disable_all()
    so.features.bursts = True
    so.postprocessing.feature_normalization = False
    notch = orc.NotchFilter(1000.0, 50)
    bu = orc.Bursts(so, names, 1000.0)
    cols = [i for i, k in enumerate(eng.keys) if "_bursts_" in k and k.split("_bursts_")[0] in names]
    keys = [eng.keys[i] for i in cols]
    for i in range(n_hops):
        w = notch.process(xr[:, starts[i]:starts[i] + 1000])
        want = bu.calc_feature(w)
        assert list(want) == keys
        if i % 10 and i < 280:
            continue   # every tenth hop while the history fills, every hop across and after the overflow
        n_bad, rep, _ = parity.compare(keys, got[i][cols], list(want.values()), so, 1000.0, 60.0, 1000,
                                       verifier=parity.Verifier(so, names, 1000.0, w, bursts=bu, raw=x[:, starts[i]:starts[i] + 1000].astype(np.float64), n_stages=2))
        assert n_bad == 0, f"hop {i}\n{rep}"


def _config3_settings():
    from py_neuromodulation_amd import NMSettings

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = s.features.stft = s.features.bursts = True
    s.frequency_ranges_hz = {"theta": [4, 8], "alpha": [8, 12], "low_beta": [13, 20], "high_beta": [20, 35],
                             "low_gamma": [60, 80], "high_gamma": [90, 200], "HFA": [200, 400],
                             "broadband": [4, 400]}
    s.bandpass_filter_settings.segment_lengths_ms["broadband"] = 1000
    return s.validate()


def test_config3_2khz_8bands_256ch(gpu_lib):
    """BASELINE config[2] at its FULL per-GPU width: 256 ch @ 2 kHz, W = 2000, hop = 200, 8-band band-pass
    bank (L = 1999), STFT(500) and bursts on two bands.  Hops 0..2 of the batch against the CPU oracle (the
    bursts history is part of the comparison: hop k needs hops 0..k), and every stateless column of the
    batch must equal the one-window call bit for bit."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd.engine import HotPathEngine

    s = _config3_settings()
    sfreq, C, n_hops, W, hop = 2000.0, 256, 5, 2000, 200
    T = W + (n_hops - 1) * hop
    rng = np.random.default_rng(5)
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + rng.uniform(-300, 300, (C, 1))).astype(np.float32)
    ch = [f"ch{i}" for i in range(C)]
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    starts = np.arange(n_hops) * hop
    got = eng.process_batch(x, starts)
    assert not np.isnan(got).any()
    eng1 = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    stateless = np.array(["_bursts_" not in k for k in eng.keys])
    for i in (0, 4):
        one = eng1.process_window(x[:, starts[i]:starts[i] + W].astype(np.float64))
        np.testing.assert_array_equal(one[stateless], got[i][stateless])
    feats = [orc.BandPower(s, ch, sfreq), orc.STFT(s, ch, sfreq), orc.Bursts(s, ch, sfreq)]
    for i in range(3):
        w = x[:, starts[i]:starts[i] + W].astype(np.float64)
        want = {}
        for f in feats:
            want.update(f.calc_feature(w))
        assert list(want) == eng.keys
        ver = parity.Verifier(s, ch, sfreq, w, bursts=parity.BurstTrace(feats[2]))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], list(want.values()), s, sfreq, 300.0, W, verifier=ver)
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()
    eng1.close()


def test_config4_shard_256_of_1024_notch_car(gpu_lib):
    """BASELINE config[3]: 1024 ch @ 1 kHz re-referenced JOINTLY (common average over all 1024 rows), notch,
    FFT + Welch + STFT + band-pass + sharp waves; one GPU computes rows 256..511 (`channel_subset`), i.e. its
    rows of the folded re-reference matrix read all 1024 input rows (structured kernel: one group sum per
    sample instead of a dense 256 x 1024 product).  Three hops against the CPU oracle run on the same 1024
    rows; batch == one-window call bit for bit; the same shard through the dense product must agree."""
    import os

    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.data_processor import DataProcessor

    s = NMSettings.get_default()
    s.features.disable_all()
    for f in ("fft", "welch", "stft", "bandpass_filter", "sharpwave_analysis"):
        setattr(s.features, f, True)
    s.preprocessing = ["notch_filter", "re_referencing"]
    s.postprocessing.feature_normalization = False
    C_all, n_hops, W, hop = 1024, 8, 1000, 100
    T = W + (n_hops - 1) * hop
    rng = np.random.default_rng(44)
    t = np.arange(T) / 1000.0
    x = (rng.standard_normal((C_all, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + 5 * np.sin(2 * np.pi * 50 * t)
         + rng.uniform(-500, 500, (C_all, 1))).astype(np.float32)
    names = [f"ch{i}" for i in range(C_all)]
    channels = {"name": names, "rereference": ["average"] * C_all, "used": [1] * C_all, "target": [0] * C_all,
                "type": ["ecog"] * C_all, "status": ["good"] * C_all, "new_name": [f"{n}_avgref" for n in names]}
    shard = range(256, 512)
    starts = np.arange(n_hops) * hop
    dp = DataProcessor(1000.0, s, channels, line_noise=50, verbose=False, lib=gpu_lib, window=W, channel_subset=shard)
    assert dp.engine.C == 256 and dp.engine.C_in == 1024
    got = dp.engine.process_batch(x, starts)
    assert "nmx_kern_reref_struct" in dp.engine.kernels(1) or "emulator" in dp.engine.kernels(1)
    assert not np.isnan(got).any()
    one = dp.engine.process_window(x[:, starts[5]:starts[5] + W].astype(np.float64))
    np.testing.assert_array_equal(one, got[5])
    os.environ["NMX_REREF_STRUCT"] = "0"
    try:
        dpd = DataProcessor(1000.0, s, channels, line_noise=50, verbose=False, lib=gpu_lib, window=W, channel_subset=shard)
        dense = dpd.engine.process_batch(x, starts[:2])
        assert ("nmx_kern_reref" in dpd.engine.kernels(1) and "struct" not in dpd.engine.kernels(1)) or "emulator" in dpd.engine.kernels(1)
    finally:
        del os.environ["NMX_REREF_STRUCT"]
    # oracle: pre-process all 1024 rows, features of the shard's rows
    odp = orc.DataProcessor(1000.0, s, channels, line_noise=50)
    sub_names = [channels["new_name"][i] for i in shard]
    feats = [orc._FEATURE_CLS[f](s, sub_names, 1000.0) for f in dp.engine.enabled]
    for i in (0, 1, 7):
        pre = odp.preprocess(x[:, starts[i]:starts[i] + W].astype(np.float64))[list(shard)]
        want = {}
        for f in feats:
            want.update(f.calc_feature(pre))
        ver = parity.Verifier(s, sub_names, 1000.0, pre, raw=x[:, starts[i]:starts[i] + W].astype(np.float64))
        n_bad, rep, _ = parity.compare(dp.engine.keys, got[i], [want[k] for k in dp.engine.keys], s, 1000.0, 700.0, W,
                                       verifier=ver)
        assert n_bad == 0, f"hop {i}\n{rep}"
        if i < 2:   # dense product of the same rows: same features within the policy
            n_bad, rep, _ = parity.compare(dp.engine.keys, dense[i], [want[k] for k in dp.engine.keys], s, 1000.0, 700.0, W,
                                           verifier=ver)
            assert n_bad == 0, f"dense hop {i}\n{rep}"
    # the same shard WITHOUT the other ranks' rows: own 256 rows + two rows (hi, lo) holding the float64 sum over all
    # 1024 (what the all-reduce of sharding.ShardedStream / bench.py --config c4 delivers)
    dpl = DataProcessor(1000.0, s, channels, line_noise=50, verbose=False, lib=gpu_lib, window=W, channel_subset=shard,
                        local_inputs=True)
    from py_neuromodulation_amd.channels import split_hi_lo

    assert dpl.local_rows == list(shard) and dpl.engine.C_in == 258
    xs = np.concatenate([x[shard.start:shard.stop], split_hi_lo(x.astype(np.float64).sum(axis=0))])
    loc = dpl.engine.process_batch(xs, starts[:2])
    assert dpl.engine.keys == dp.engine.keys
    for i in (0, 1):
        pre = odp.preprocess(x[:, starts[i]:starts[i] + W].astype(np.float64))[list(shard)]
        want = {}
        for f in feats:
            want.update(f.calc_feature(pre))
        ver = parity.Verifier(s, sub_names, 1000.0, pre, raw=x[:, starts[i]:starts[i] + W].astype(np.float64))
        n_bad, rep, _ = parity.compare(dp.engine.keys, loc[i], [want[k] for k in dp.engine.keys], s, 1000.0, 700.0, W, verifier=ver)
        assert n_bad == 0, f"local-input hop {i}\n{rep}"
    dp.engine.close()
    dpd.engine.close()
    dpl.engine.close()


def test_config5_30khz_512ch(gpu_lib):
    """BASELINE config[4] at its FULL per-GPU width: 512 of 4096 ch @ 30 kHz, 512-sample windows, hop 30
    (settings of tests/parity_cases.case_config5_30khz_512pt), three hops vs the oracle + batch == one-window."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    base = NMSettings.get_default().to_dict()
    base["frequency_ranges_hz"] = {"gamma": [60, 200], "HFA": [200, 500], "MUA": [500, 3000], "spike": [3000, 7000]}
    s = NMSettings(**base)
    s.features.disable_all()
    for f in ("fft", "stft", "raw_hjorth", "linelength", "return_raw", "bandpass_filter", "sharpwave_analysis"):
        setattr(s.features, f, True)
    s.sampling_rate_features_hz = 1000
    s.segment_length_features_ms = 17
    s.fft_settings.windowlength_ms = 17
    s.stft_settings.windowlength_ms = 17
    s.bandpass_filter_settings.segment_lengths_ms = {"gamma": 17, "HFA": 10, "MUA": 5, "spike": 3}
    s.sharpwave_analysis_settings.filter_ranges_hz = [[500, 3000], [1000, 7000]]
    s = NMSettings(**s.to_dict())
    sfreq, C, W, hop, nh = 30000.0, 512, 512, 30, 16
    rng = np.random.default_rng(0)
    T = W + (nh - 1) * hop
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 30 + 40 * np.sin(2 * np.pi * 900 * t) + rng.uniform(-100, 100, (C, 1))).astype(np.float32)
    ch = [f"c{i}" for i in range(C)]
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib, window=W)
    starts = np.arange(nh) * hop
    got = eng.process_batch(x, starts)
    one = eng.process_window(x[:, starts[9]:starts[9] + W].astype(np.float64))
    np.testing.assert_array_equal(one, got[9])
    feats = [orc._FEATURE_CLS[f](s, ch, sfreq) for f in eng.enabled]
    for i in (0, 7, 15):
        w = x[:, starts[i]:starts[i] + W].astype(np.float64)
        want = {}
        for f in feats:
            want.update(f.calc_feature(w))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], [want[k] for k in eng.keys], s, sfreq, 200.0, W,
                                       verifier=parity.Verifier(s, ch, sfreq, w))
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()


def test_ragged_float_sfreq_stream(gpu_lib):
    pc.case_ragged_float_sfreq_stream(gpu_lib)


def test_ragged_windows_carry_burst_and_kalman_state(gpu_lib):
    pc.case_ragged_bursts(gpu_lib)


def test_long_windows_sharp_waves_and_order_normalisers(gpu_lib):
    pc.case_long_windows(gpu_lib)


def test_raw_normalisation_with_ragged_window_lengths(gpu_lib):
    pc.case_ragged_rawnorm(gpu_lib)


def test_raw_order_normalisers_with_lists_in_device_memory(gpu_lib, monkeypatch):
    """The reference goldens of the order-statistic raw normalisers (1000-sample windows) with the merge lists forced
    into device memory -- the layout windows beyond 6484 samples take."""
    monkeypatch.setenv("NMX_RAWNORM_GLOBAL_LISTS", "1")
    pc.case_raw_normalizer_order_methods(gpu_lib)


def test_odd_windows_and_spectra(gpu_lib):
    pc.case_odd_windows_and_spectra(gpu_lib)


def test_reference_property_tests(gpu_lib):
    """Drop-in plugin classes (package loader, no lib injection) under the reference's property tests."""
    pc.case_reference_property_tests(gpu_lib)


def test_feature_normalizer_batches(gpu_lib):
    pc.case_feature_normalizer_batches(gpu_lib)


def test_feature_normalizer_power(gpu_lib):
    pc.case_feature_normalizer_power(gpu_lib)


def test_raw_quantile_subsample(gpu_lib):
    pc.case_raw_quantile_subsample(gpu_lib)


def test_resampler_long_windows(gpu_lib):
    pc.case_resampler_long_windows(gpu_lib)


def test_high_rate_partitioned_fir(gpu_lib):
    pc.case_high_rate_partitioned_fir(gpu_lib)


@pytest.mark.parametrize("seed", list(range(8)) + [102790])   # (102790: an edge transient in x[0] behind the resampler -- the matrix-pipe kernel's pivot)
def test_random_settings_highrate(gpu_lib, seed):
    pc.case_random_settings_highrate(gpu_lib, seed)


def test_config5_degenerate_bursts_and_welch(gpu_lib):
    pc.case_config5_degenerate(gpu_lib)


def test_bandpower_kalman_sequence(gpu_lib):
    pc.case_bandpower_kalman_sequence(gpu_lib)


def test_resampler(gpu_lib):
    pc.case_resampler(gpu_lib)


def test_preprocessing_filter(gpu_lib):
    pc.case_preprocessing_filter(gpu_lib)


def test_config5_30khz_512pt(gpu_lib):
    pc.case_config5_30khz_512pt(gpu_lib)


def test_raw_normalizer(gpu_lib):
    pc.case_raw_normalizer(gpu_lib)


def test_raw_normalizer_order_methods(gpu_lib):
    pc.case_raw_normalizer_order_methods(gpu_lib)


def test_psd_keys_skip_normalisation(gpu_lib):
    pc.case_psd_keys_skip_normalisation(gpu_lib)


def test_reref_structured_matrices(gpu_lib):
    pc.case_reref_structured_matrices(gpu_lib)


def test_raw_resampling_reference_quirk(gpu_lib):
    pc.case_raw_resampling_reference_quirk(gpu_lib)


def test_alternative_code_paths_agree(gpu_lib, monkeypatch):
    """Plan-level knobs select fallback / alternative kernels (list-based sharp-wave code, dense
    re-reference, serial launch order, fused sharp waves, fused Hilbert envelopes, global-memory burst list,
    block-wide STFT, generic time / oscillatory kernel, small chunks).  EVERY path -- the default one
    included -- is compared with the float64 oracle over the whole 72-hop stream under the standard policy
    (bursts history included), so a path cannot hide behind another path's rounding."""
    from oracle import nm_oracle as orc

    C, n_hops = 64, 72    # 4608 items: enough for the persistent bank kernel (>= 4096) and its fused variants
    T = 1000 + (n_hops - 1) * 100
    rng = np.random.default_rng(77)
    t = np.arange(T) / 1000.0
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + rng.uniform(-300, 300, (C, 1))).astype(np.float32)
    starts = np.arange(n_hops) * 100

    def run():
        s, eng = _bench_like_engine(gpu_lib, C)
        out = eng.process_batch(x, starts)
        keys = list(eng.keys)
        eng.close()
        return s, keys, out

    s, keys, default = run()
    names = [f"ch{i}" for i in range(C)]
    channels = {"name": names, "rereference": ["average"] * C, "used": [1] * C, "target": [0] * C,
                "type": ["ecog"] * C, "status": ["good"] * C, "new_name": [f"{n}_avgref" for n in names]}
    so = type(s)(**s.to_dict())
    so.postprocessing.feature_normalization = False
    so.preprocessing = ["notch_filter", "re_referencing"]
    dp = orc.DataProcessor(1000.0, so, channels, line_noise=50)
    want = []
    for a in starts:
        d = dp.process(x[:, a:a + 1000].astype(np.float64))
        want.append([d[k] for k in keys])
    pv = parity.PipelineVerifiers(so, channels, 1000.0, x, starts, 1000, line_noise=50)

    def check(tag, got):
        for i in range(n_hops):
            b, rep, _ = parity.compare(keys, got[i], want[i], so, 1000.0, 400.0, 1000, verifier=pv.row(i))
            assert b == 0, f"{tag} hop {i}\n{rep}"

    check("default", default)
    # the code paths a SHAPE can select that this shape does not reach by itself: every filter on the M = 2048
    # channel-pair / one-channel kernels, the one-channel notch, the generic sharp-wave / time-oscillatory / threshold-walk kernels, the schedules, a
    # chunk boundary every 9 hops
    for knobs in ({"NMX_BANK_W64C": "0"}, {"NMX_BANK_W64E": "0"}, {"NMX_BANK_W64C": "0", "NMX_BANK_W64E": "0"}, {"NMX_SW_DENSE": "0"}, {"NMX_CAR_FAST": "0"}, {"NMX_OVERLAP": "0"},
                  {"NMX_OVERLAP": "2"}, {"NMX_THR_LIST_GLOBAL": "1"}, {"NMX_THR_FILL": "0"}, {"NMX_FILL_SPLIT": "0"}, {"NMX_TIMEOSC_W1000": "0"},
                  {"NMX_TOW_PERSISTENT": "0"}, {"NMX_CHUNK_WINDOWS": "9"}, {"NMX_WAVES_PER_WG": "1"}, {"NMX_WAVES_PER_WG": "3"},
                  {"NMX_NOTCH_RESIDUAL": "0"}, {"NMX_NOTCH_RESIDUAL": "0", "NMX_BANK_W64E": "0"}):
        for knob, val in knobs.items():
            monkeypatch.setenv(knob, val)
        _, keys2, got = run()
        for knob in knobs:
            monkeypatch.delenv(knob)
        assert keys2 == keys
        check(" ".join(f"{k}={v}" for k, v in knobs.items()), got)


def test_channel_pair_bank_odd_count_and_unequal_scales(gpu_lib):
    """The M = 1536 FIR-bank kernel carries two channels in one complex transform (nmx_k_bank_w64c.h).  An ODD
    channel count leaves the last channel alone in its transform; neighbours whose amplitudes differ by 10^4 and
    10^-3 must each keep the accuracy they would have alone (the second channel is rescaled by a power of two);
    a flat channel next to a live one stays exactly flat.  Against the float64 oracle under the standard policy,
    and batch == window-by-window bit for bit (the pairing does not depend on how hops are batched)."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    C, n_hops, W = 7, 6, 1000
    T = W + (n_hops - 1) * 100
    rng = np.random.default_rng(123)
    t = np.arange(T) / 1000.0
    x = rng.standard_normal((C, T)) * 30 + 8 * np.sin(2 * np.pi * 21 * t)
    x[1] *= 1e4       # loud second half of pair (0, 1)
    x[2] *= 1e4       # loud first half of pair (2, 3)
    x[5] *= 1e-3      # quiet second half of pair (4, 5)
    x[4] = 0.0        # flat first half
    x = x.astype(np.float32)
    starts = np.arange(n_hops) * 100
    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = True
    names = [f"ch{i}" for i in range(C)]
    eng = HotPathEngine(s, names, 1000.0, lib=gpu_lib)
    got = eng.process_batch(x, starts)
    assert "w64c" in eng.kernels(3)
    keys = list(eng.keys)
    one = np.stack([eng.process_batch(x[:, a:a + W].copy(), np.zeros(1, dtype=np.int64))[0] for a in starts])
    np.testing.assert_array_equal(got, one)
    eng.close()
    feats = [orc._FEATURE_CLS[n](s, names, 1000.0) for n in s.features.get_enabled()]
    for i, a in enumerate(starts):
        want: dict = {}
        for f in feats:
            want.update(f.calc_feature(x[:, a:a + W].astype(np.float64)))
        assert list(want) == keys
        ver = parity.Verifier(s, names, 1000.0, x[:, a:a + W].astype(np.float64))
        b, rep, _ = parity.compare(keys, got[i], [want[k] for k in keys], s, 1000.0, 400.0, W, verifier=ver)
        assert b == 0, f"hop {i}\n{rep}"


def test_linearity_of_the_filter_stages(gpu_lib):
    """Size-independent property at the headline width (256 ch): notch + re-reference and the FIR bank
    are linear maps of the window: T(a x + b y) = a T(x) + b T(y) within fp32 rounding of the sums."""
    from py_neuromodulation_amd import NMSettings, fir_design
    from py_neuromodulation_amd.engine import HotPathEngine

    C, W = 256, 1000
    rng = np.random.default_rng(5)
    x = rng.standard_normal((C, W)) * 40 + rng.uniform(-200, 200, (C, 1))
    y = rng.standard_normal((C, W)) * 25
    a, b = 0.75, -1.5
    R = np.full((C, C), -1.0 / (C - 1))
    np.fill_diagonal(R, 1.0)
    s = NMSettings.get_default()
    s.features.bandpass_filter = True
    eng = HotPathEngine(s, [f"ch{i}" for i in range(C)], 1000.0, lib=gpu_lib, ref_matrix=R,
                        notch_taps=fir_design.notch_bank(1000.0, 50))
    px, py_, pz = eng.preprocess_window(x), eng.preprocess_window(y), eng.preprocess_window(a * x + b * y)
    scale = np.abs(px).max() + np.abs(py_).max()
    np.testing.assert_allclose(pz, a * px + b * py_, rtol=0, atol=3e-6 * scale)
    fx, fy, fz = eng.filter_window(px), eng.filter_window(py_), eng.filter_window(a * px + b * py_)
    assert fx.shape == (C, int(eng.desc.n_filters), W)
    np.testing.assert_allclose(fz, a * fx + b * fy, rtol=0, atol=3e-6 * scale)
    eng.close()


@pytest.mark.parametrize("sfreq,ring_s,n_hops", [(1000.0, 30, 900), (2000.0, 3, 400)])
def test_threshold_walk_one_wave_equals_workgroup_kernel(gpu_lib, monkeypatch, sfreq, ring_s, n_hops):
    """The burst threshold walk has two implementations: the 256-thread workgroup kernel (fill regime and
    transitions) and the barrier-free one-wave kernel used when the 30 s ring is already full at the first
    hop of a batch.  Both move the same values around, so every burst feature must agree BIT FOR BIT over a
    long stream (default ring: 291 fill hops, then steady), fed in chunks of 128 hops, in uneven batches and
    with a state export / import in the middle."""
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.bursts_settings.time_duration_s = ring_s   # (2 kHz: 200 new samples per hop -> four registers per lane in the wave kernel)
    s = s.validate()
    C, W, hop = 3, int(sfreq), int(sfreq / 10)
    T = W + (n_hops - 1) * hop
    rng = np.random.default_rng(5)
    t = np.arange(T) / sfreq
    amp = 1 + 0.8 * np.sin(2 * np.pi * 0.05 * t) + 2 * t / t[-1]          # slowly growing power: steady inserts
    data = (rng.standard_normal((C, T)) * 20 + 30 * amp * np.sin(2 * np.pi * 18 * t)).astype(np.float32)
    data[1] *= 1e-3                                                        # tiny amplitudes
    data[2] = np.round(data[2])                                            # many equal values (ties)
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * hop
    monkeypatch.setenv("NMX_CHUNK_WINDOWS", "128")

    def run(wave, plan, export_at=None, fill=True):
        monkeypatch.setenv("NMX_THR_WAVE", "1" if wave else "0")
        monkeypatch.setenv("NMX_THR_FILL", "1" if fill else "0")   # fresh stream: the sort-once fill walk (nmx_k_burst_fill.h)
        eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib, features=["bursts"], bank_taps=None)
        rows, i = [], 0
        for n in plan:
            if export_at is not None and i == export_at:
                blob = eng.export_state()
                eng.close()
                eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib, features=["bursts"], bank_taps=None)
                eng.import_state(blob)
            rows.append(eng.process_batch(data, starts[i:i + n]))
            i += n
        eng.close()
        return np.concatenate(rows)

    want = run(False, [n_hops], fill=False)          # the workgroup kernel alone
    assert not np.isnan(want).any()
    np.testing.assert_array_equal(run(False, [n_hops]), want)            # fill walk + workgroup kernel
    np.testing.assert_array_equal(run(True, [n_hops], fill=False), want)
    np.testing.assert_array_equal(run(True, [n_hops]), want)             # the default: fill walk, then the one-wave walk
    monkeypatch.setenv("NMX_FILL_SPLIT", "0")                              # the fill as ONE launch (sort and walk in one workgroup)
    np.testing.assert_array_equal(run(True, [n_hops]), want)
    np.testing.assert_array_equal(run(False, [n_hops]), want)
    monkeypatch.delenv("NMX_FILL_SPLIT")
    np.testing.assert_array_equal(run(True, [5, 1, n_hops - 6]), want)   # a fill walk of 5 hops, continued by the others
    monkeypatch.setenv("NMX_CHUNK_WINDOWS", "1024")
    np.testing.assert_array_equal(run(True, [n_hops]), want)             # one chunk: the fill walk ends where the ring is full
    monkeypatch.setenv("NMX_CHUNK_WINDOWS", "128")
    a, b = n_hops // 3, n_hops // 2
    np.testing.assert_array_equal(run(True, [a, 1, 7, n_hops - 2 * a - 8, a]), want)
    np.testing.assert_array_equal(run(True, [b, n_hops - b], export_at=b), want)


@pytest.mark.parametrize("W", [800, 901])
def test_persistent_bank_equals_one_window_kernel_other_lengths(gpu_lib, W):
    """Batches of >= 4096 (window, channel) items run the persistent FIR-bank kernel (tables in LDS, lower
    half of the inverse transforms only); a single window runs the one-wave-per-workgroup kernel.  Same
    arithmetic, so band-pass power, sharp-wave and (first-hop) burst features must agree bit for bit --
    here for an even and an odd window length other than the default 1000 (odd: dword-wise buffer
    accesses), and against the CPU oracle on two hops."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = s.features.sharpwave_analysis = True
    s.segment_length_features_ms = W
    s.bandpass_filter_settings.segment_lengths_ms = {"theta": W, "alpha": 500, "low_beta": 333, "high_beta": 333}
    s = s.validate()
    sfreq, C, n_hops = 1000.0, 64, 66
    T = W + (n_hops - 1) * 100
    rng = np.random.default_rng(W)
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 50 + 10 * np.sin(2 * np.pi * 20 * t) + rng.uniform(-300, 300, (C, 1))).astype(np.float32)
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * 100
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    got = eng.process_batch(x, starts)          # 66 * 64 = 4224 items: persistent kernel
    assert not np.isnan(got).any()
    for i in (0, 31, 65):
        one = eng.process_window(x[:, starts[i]:starts[i] + W].astype(np.float64))
        np.testing.assert_array_equal(one, got[i])
    feats = [orc._FEATURE_CLS[f](s, ch, sfreq) for f in eng.enabled]
    for i in (3, 40):
        want = {}
        for f in feats:
            want.update(f.calc_feature(x[:, starts[i]:starts[i] + W].astype(np.float64)))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], [want[k] for k in eng.keys], s, sfreq, 400.0, W,
                                       verifier=parity.Verifier(s, ch, sfreq, x[:, starts[i]:starts[i] + W].astype(np.float64)))
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()


@pytest.mark.parametrize("sfreq,kernel", [(500.0, "w64d_rd64<1>"), (600.0, "w64d_rd64<0>"), (750.0, "w64c")])
def test_channel_pair_bank_other_rates(gpu_lib, sfreq, kernel):
    """One-second windows at 500 / 600 / 750 Hz: the band-pass taps (sfreq - 1 long) need M >= 1.5 W -- 749 and 899
    fit the 1024-point channel-pair kernel (windows <= 512: only half of the inverse outputs are formed; > 512: all
    of them), 1124 the 1536-point one.  An odd channel count, batch == window by window bit for bit, two hops against
    the float64 oracle."""
    from oracle import nm_oracle as orc
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.engine import HotPathEngine

    s = NMSettings.get_default()
    s.features.disable_all()
    s.features.bandpass_filter = True
    s = s.validate()
    W, hop = int(sfreq), int(sfreq / 10)
    C, n_hops = 33, 40
    T = W + (n_hops - 1) * hop
    rng = np.random.default_rng(int(sfreq))
    t = np.arange(T) / sfreq
    x = (rng.standard_normal((C, T)) * 40 + 12 * np.sin(2 * np.pi * 18 * t) + rng.uniform(-200, 200, (C, 1))).astype(np.float32)
    ch = [f"ch{i}" for i in range(C)]
    starts = np.arange(n_hops) * hop
    eng = HotPathEngine(s, ch, sfreq, lib=gpu_lib)
    got = eng.process_batch(x, starts)
    assert kernel in eng.kernels(3), eng.kernels(3)
    for i in (0, 17, 39):
        one = eng.process_window(x[:, starts[i]:starts[i] + W].astype(np.float64))
        np.testing.assert_array_equal(one, got[i])
    feats = [orc._FEATURE_CLS[f](s, ch, sfreq) for f in eng.enabled]
    for i in (5, 30):
        want = {}
        for f in feats:
            want.update(f.calc_feature(x[:, starts[i]:starts[i] + W].astype(np.float64)))
        n_bad, rep, _ = parity.compare(eng.keys, got[i], [want[k] for k in eng.keys], s, sfreq, 200.0, W,
                                       verifier=parity.Verifier(s, ch, sfreq, x[:, starts[i]:starts[i] + W].astype(np.float64)))
        assert n_bad == 0, f"hop {i}\n{rep}"
    eng.close()


def test_windows_shorter_than_the_spectral_segment(gpu_lib):
    pc.case_short_windows(gpu_lib)


@pytest.mark.parametrize("seed", pc.RANDOM_SETTINGS_SEEDS)
def test_random_settings_stream_equals_oracle(gpu_lib, seed):
    pc.case_random_settings(gpu_lib, seed)


@pytest.mark.parametrize("seed", pc.WIDE_SETTINGS_SEEDS)
def test_random_settings_wide_stream_equals_oracle(gpu_lib, seed):
    pc.case_random_settings_wide(gpu_lib, seed)


@pytest.mark.parametrize("seed", pc.CHANNEL_TABLE_SEEDS)
def test_random_channel_tables_stream_equals_oracle(gpu_lib, seed):
    pc.case_random_channel_tables(gpu_lib, seed)


@pytest.mark.parametrize("seed", pc.BURST_STREAM_SEEDS)
def test_random_burst_streams_equal_oracle(gpu_lib, seed):
    pc.case_random_burst_streams(gpu_lib, seed)


@pytest.mark.parametrize("seed", pc.RANDOM_SETTINGS_SEEDS[:20])
def test_random_settings_window_by_window_equals_batch(gpu_lib, seed):
    pc.case_random_window_by_window(gpu_lib, seed)


def test_user_registered_features(gpu_lib):
    pc.case_user_features(gpu_lib)


def test_stream_output_files(gpu_lib, tmp_path):
    pc.case_stream_output_files(gpu_lib, tmp_path)


@pytest.mark.parametrize("large", [False, True])
def test_input_layouts_give_identical_results(gpu_lib, large):
    pc.case_input_layouts(gpu_lib, large=large)


def test_input_layouts_single_channel(gpu_lib):
    pc.case_input_layouts_single_channel(gpu_lib)


@pytest.mark.parametrize("devices", [None, (0, 0)])
def test_real_recording_of_the_reference_tests(gpu_lib, devices):
    pc.case_real_recording(gpu_lib, devices=devices)


@pytest.mark.parametrize("tag", ["", "stft_"])
def test_reref_group_members_on_the_rail(gpu_lib, tag):
    pc.case_inf_members(gpu_lib, tag=tag, spectral_nan_ok=False)


def test_trends_are_counted_not_hidden(gpu_lib):
    acc = pc.case_trends(gpu_lib)
    assert sum(acc.values()) <= 40, acc   # (emulator 11, all Welch bins at 1e-4 of the swell's leakage; see the budget file)


def test_watchdog_names_the_kernels_of_a_batch_that_does_not_finish(tmp_path):
    """hipStreamSynchronize has no timeout; the wait at the end of a host-memory batch polls with one
    (NMX_SYNC_TIMEOUT_S, nmx_api.hip: be_sync_watch) and its error lists the launch sequence stage by stage.  Shown with
    a limit far below a 1024-hop batch's run time, in a process of its own (the limit is read once)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "import bench\n"
        "from py_neuromodulation_amd import fir_design, _lib\n"
        "from py_neuromodulation_amd.engine import HotPathEngine\n"
        "s = bench.make_settings(); C = 256\n"
        "eng = HotPathEngine(s, [f'ch{i}' for i in range(C)], 1000.0, ref_matrix=bench.car_matrix(C), notch_taps=fir_design.notch_bank(1000.0, 50))\n"
        "x = eng.pinned_empty((C, 1000 + 1023 * 100)); x[...] = bench.synth(C, 1000 + 1023 * 100, 1000.0, 1)\n"
        "out = eng.pinned_empty((1024, eng.n_outputs))\n"
        "try:\n"
        "    eng.process_batch(x, np.arange(1024) * 100, out=out)   # (page-locked both ways: ~6 ms of device work queued in < 1 ms)\n"
        "except _lib.NmxError as e:\n"
        "    print('CAUGHT', e)\n"
    )
    env = dict(__import__("os").environ, NMX_SYNC_TIMEOUT_S="0.00001")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert "CAUGHT" in res.stdout and "did not finish" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
    assert "nmx_kern_bank_w64c" in res.stdout and "stage 4" in res.stdout, res.stdout[-2000:]


def test_nan_on_an_offset_channel_without_a_rereference(gpu_lib):
    pc.case_dc_nan(gpu_lib)


def test_abi_from_plain_c_on_the_gpu(tmp_path):
    """tests/c_abi/abi_smoke.c on a box WITH a device: its `ndev > 0` branch creates a plan and computes one
    feature through the C ABI from plain C (the CPU tier only reaches the argument checks)."""
    import subprocess
    from pathlib import Path

    import __graft_entry__ as g

    root = Path(__file__).resolve().parent.parent
    lib = g.build_lib()
    exe = tmp_path / "abi_smoke"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", str(root / "tests" / "c_abi" / "abi_smoke.c"), "-I", str(root / "include"),
           "-L", str(lib.parent), "-lnmx", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK devices="), r.stdout + r.stderr
    assert int(r.stdout.split("=")[1]) >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_gloo_ranks_share_one_gpu(scaling):
    """bench.py under the driver's launcher with two ranks on the one GPU of the box (gloo): the weak form (256 channels
    per rank, no collective) and the strong form (256 channels in TOTAL, jointly re-referenced: one all-reduce of the
    column sums per step) both print ONE line that says which it is."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NMX_BENCH_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "128",
           "--cpu-windows", "0", "--no-cold-start", "--no-mode-a", "--backend", "gloo", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(root))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["nan_outputs"] == 0
    assert len(d["ms_per_step_by_rank"]) == 2
    if scaling == "strong":
        assert d["config"]["channels_total"] == 256 and d["config"]["channels_per_gpu"] == 128
        assert d["exchange_ms_per_step"] is not None
    else:
        assert d["config"]["channels_per_gpu"] == 256


def test_standalone_classes_any_length(gpu_lib):
    pc.case_standalone_classes_any_length(gpu_lib)


def test_plugin_classes_as_the_reference_uses_them(gpu_lib):
    pc.case_plugin_classes_as_the_reference_uses_them(gpu_lib)


def test_standalone_rereferencer_float64(gpu_lib):
    pc.case_standalone_rereferencer_float64(gpu_lib)


def test_standalone_resampler_float64(gpu_lib):
    pc.case_standalone_resampler_float64(gpu_lib)


def test_processor_hop_by_hop_with_two_window_lengths(gpu_lib):
    pc.case_processor_hop_by_hop_with_two_window_lengths(gpu_lib)

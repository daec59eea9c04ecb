"""Multi-process (world_size 2, gloo, CPU) test of the channel-sharded path: two ranks compute
disjoint channel blocks through the kernel-logic emulator and rank 0 assembles the table; it
must equal the single-process result column for column (incl. common-average re-referencing,
whose rows read all input channels)."""

import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import __graft_entry__ as ge
from py_neuromodulation_amd import NMSettings, _lib
from py_neuromodulation_amd import channels as chmod
from py_neuromodulation_amd.sharding import ShardedStream, gather_dataframe, global_keys, channel_shard

backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
rank_env = int(os.environ["LOCAL_RANK"])
if backend == "nccl":          # the real target: RCCL, one GPU per rank, the product library
    import torch
    torch.cuda.set_device(rank_env)
    lib, dev = _lib.get_library(), rank_env
else:                          # CPU tier: gloo + the kernel-logic emulator
    lib, dev = _lib.NmxLibrary(ge.build_emu()), 0
dist.init_process_group(backend)
rank, world = dist.get_rank(), dist.get_world_size()
s = NMSettings.get_default()
s.features.bandpass_filter = True
s.preprocessing = ["notch_filter", "re_referencing"]
s.postprocessing.feature_normalization = False
rng = np.random.default_rng(7)
data = rng.standard_normal((5, 2500)) * 20 + rng.uniform(-100, 100, (5, 1))
ch = chmod.get_default_channels_from_data(data)
st = ShardedStream(1000.0, ch, s, line_noise=50, rank=rank, world_size=world, device=dev, lib=lib)
keys, rows, times = st.run(data)
assert len(keys) == len(set(keys))
allk = global_keys(1000.0, s, ch)
df = gather_dataframe(keys, rows, times, allk)
if rank == 0:
    df.to_pickle(sys.argv[2])
dist.barrier()
# local input: each rank is handed ONLY the rows of its own channels (+ the rows its bipolar references
# name); the type-group sums of the "average" references travel through ONE all-reduce, the NaN mask
# through an all-gather.  Mixed table: ecog average group, an lfp pair referenced to named channels.
ch2 = ch.copy()
ch2.loc[3, "type"] = "lfp"; ch2.loc[3, "rereference"] = "ch0"; ch2.loc[3, "new_name"] = "ch3_ch0"
ch2.loc[4, "type"] = "lfp"; ch2.loc[4, "rereference"] = "ch3&ch1"; ch2.loc[4, "new_name"] = "ch4_ch3ch1"
data2 = data.copy()
data2[1, 1200:1210] = np.nan
st2 = ShardedStream(1000.0, ch2, s, line_noise=50, rank=rank, world_size=world, device=dev, lib=lib, local_input=True)
assert set(st2.owned_rows) <= set(st2.local_rows) and (world == 1 or len(st2.local_rows) < 5)
keys2, rows2, times2 = st2.run(data2[st2.local_rows])
df2 = gather_dataframe(keys2, rows2, times2, global_keys(1000.0, s, ch2))
if rank == 0:
    df2.to_pickle(sys.argv[2] + ".local")
dist.barrier()
# ragged window lengths (1111.111 Hz: windows of 1111 and 1112 samples): one processor per length on every rank, the
# rank's burst histories and Kalman filters handed over where the length changes -- both input forms
s3 = NMSettings.get_default()
s3.reset()
s3.features.fft = True
s3.features.bursts = True
s3.features.bandpass_filter = True
s3.bandpass_filter_settings.kalman_filter = True
s3.kalman_filter_settings.frequency_bands = ["theta", "low_beta"]
s3.bursts_settings.time_duration_s = 2
s3.preprocessing = ["re_referencing"]
s3.postprocessing.feature_normalization = True
rng3 = np.random.default_rng(19)
t3 = np.arange(4500) / 1111.111
data3 = rng3.standard_normal((5, 4500)) * 20 + 15 * np.sin(2 * np.pi * 18 * t3) * (np.sin(2 * np.pi * 0.7 * t3) > 0)
data3[2, 2000:2003] = np.nan
for tag, local in (("ragged", False), ("ragged_local", True)):
    st3 = ShardedStream(1111.111, ch, s3, line_noise=50, rank=rank, world_size=world, device=dev, lib=lib, local_input=local)
    keys3, rows3, times3 = st3.run(data3[st3.local_rows] if local else data3)
    df3 = gather_dataframe(keys3, rows3, times3, global_keys(1111.111, s3, ch))
    if rank == 0:
        df3.to_pickle(sys.argv[2] + "." + tag)
    dist.barrier()
# several members of the common-average group on the rail in one sample (tests/golden/inf_members.npz, the reference's own
# run): the all-reduced float64 group sum must carry 2 x FLT_MAX like the one-plan kernel's
from tests.helpers import load_golden, settings_from_json
g4 = load_golden("inf_members")
s4 = settings_from_json(g4["settings_json"])
ch4 = json.loads(str(g4["channels_json"]))
for tag, local in (("inf", False), ("inf_local", True)):
    st4 = ShardedStream(1000.0, ch4, s4, line_noise=50, rank=rank, world_size=world, device=dev, lib=lib, local_input=local)
    keys4, rows4, times4 = st4.run(g4["data"][st4.local_rows] if local else g4["data"])
    df4 = gather_dataframe(keys4, rows4, times4, global_keys(1000.0, s4, ch4))
    if rank == 0:
        df4.to_pickle(sys.argv[2] + "." + tag)
    dist.barrier()
dist.destroy_process_group()
'''


def _device_count() -> int:
    try:
        from py_neuromodulation_amd import _lib

        return _lib.get_library().device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="the RCCL path needs two GPUs (one rank per GPU)")
def test_two_rank_channel_shards_nccl(tmp_path):
    """The SAME worker with backend="nccl" (= RCCL) on two GPUs and the product library: the group-sum all-reduce
    and the mask all-gather run on device tensors."""
    _run_two_ranks(tmp_path, "nccl", 29519)


@pytest.mark.gpu
def test_one_rank_rccl_collectives_on_the_gpu(tmp_path):
    """What a 1-GPU box can show of the RCCL path: the same worker as ONE rank under backend="nccl" with
    NMX_FORCE_COLLECTIVES=1 -- the group-sum all-reduce runs through RCCL on a device tensor, the NaN mask and the
    feature table through the object collectives staged on the GPU -- against the single-process stream."""
    _run_two_ranks(tmp_path, "nccl", 29521, nproc=1)


def test_two_rank_channel_shards_equal_single_process(tmp_path):
    _run_two_ranks(tmp_path, "gloo", 29517)


@pytest.mark.gpu
def test_single_process_multi_device_stream_on_the_gpu():
    """The same on the product library: distinct GPUs when the box has them, else two plans on GPU 0 driven by two
    host threads at once (what a 1-GPU box can show: the plans, staging pools and streams do not interfere)."""
    from py_neuromodulation_amd import _lib

    lib = _lib.get_library()
    # fp32: the channel-pair FIR kernel pairs channel 3 with 2 in the one-plan stream and with 4 in the shard; the
    # z-score divides the 1e-6 relative difference of a log power by a standard deviation 20x smaller than its mean
    _multi_device_case(lib, [0, 1] if lib.device_count() >= 2 else [0, 0], tol=2e-4)


def test_single_process_multi_device_stream_equals_single_device():
    """Stream(devices=[...]): one plan + one host thread per device inside one process.  CPU tier: three
    plans of the kernel-logic emulator; the merged table must equal the one-plan stream column for column,
    including a z-score normaliser (per column, so sharding it is exact) and the NaN policy."""
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge

    from py_neuromodulation_amd import _lib

    _multi_device_case(_lib.NmxLibrary(ge.build_emu()), [0, 0, 0])


def _multi_device_case(lib, devices, tol=1e-6):
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.stream import Stream

    s = NMSettings.get_default()
    s.features.bandpass_filter = True
    s.preprocessing = ["notch_filter", "re_referencing"]
    rng = np.random.default_rng(11)
    data = rng.standard_normal((5, 2300)) * 20 + rng.uniform(-100, 100, (5, 1))
    data[2, 1500:1504] = np.nan
    os.environ["NMX_CAR_FAST"] = "0"   # same arithmetic on both sides (see below)
    try:
        one = Stream(1000.0, data=data, settings=s, line_noise=50, lib=lib).run(save_csv=False)
        st = Stream(1000.0, data=data, settings=s, line_noise=50, lib=lib, devices=devices)
        many = st.run(save_csv=False)
        again = st.run(save_csv=False)   # the processors are reused with their state reset
    finally:
        del os.environ["NMX_CAR_FAST"]
    assert list(many.columns) == list(one.columns)
    a, b = many.to_numpy(float), one.to_numpy(float)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a).any()
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol, equal_nan=True)
    np.testing.assert_array_equal(again.to_numpy(float), a)
    row = st.data_processor.process(np.nan_to_num(data[:, :1000]))
    assert list(row) == list(one.columns)[:-1]


def _run_two_ranks(tmp_path, backend, port, nproc=2):
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    import pandas as pd

    from py_neuromodulation_amd import NMSettings, _lib
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.sharding import channel_shard
    from py_neuromodulation_amd.stream import Stream

    assert [list(channel_shard(5, 2, r)) for r in range(2)] == [[0, 1, 2], [3, 4]]   # (the worker asserts its own shard)
    assert sum(len(channel_shard(4096, 8, r)) for r in range(8)) == 4096
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = tmp_path / "sharded.pkl"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc == 1:
        env["NMX_FORCE_COLLECTIVES"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(worker), str(ROOT), str(out), backend]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    sharded = pd.read_pickle(out)

    lib = _lib.get_library() if backend == "nccl" else _lib.NmxLibrary(ge.build_emu())
    s = NMSettings.get_default()
    s.features.bandpass_filter = True
    s.preprocessing = ["notch_filter", "re_referencing"]
    s.postprocessing.feature_normalization = False
    rng = np.random.default_rng(7)
    data = rng.standard_normal((5, 2500)) * 20 + rng.uniform(-100, 100, (5, 1))
    # same arithmetic on both sides: dense re-reference rows (the single-GPU common-average fast
    # path sums in a different order, which would only test fp32 rounding, not the sharding)
    os.environ["NMX_CAR_FAST"] = "0"
    try:
        single = Stream(1000.0, data=data, settings=s, line_noise=50, lib=lib).run(save_csv=False)
    finally:
        del os.environ["NMX_CAR_FAST"]
    assert list(sharded.columns) == list(single.columns)
    a, b = sharded.to_numpy(float), single.to_numpy(float)
    assert a.shape == b.shape and not np.isnan(a).any()
    # same kernels, same inputs per channel -> identical up to fp32 summation order in the
    # re-reference rows (identical here: each row is computed by the same code path)
    # (on the GPU the shard's structured re-reference kernel and the dense one sum in different orders and the
    # channel-pair FIR kernel pairs other channels: fp32 against fp32, 1e-3 of a feature's scale)
    loose = backend == "nccl"
    np.testing.assert_allclose(a, b, rtol=1e-3 if loose else 1e-6, atol=1e-4 if loose else 1e-6)

    # local-input mode (group sums all-reduced, masks all-gathered) == the single-process stream on the same
    # table, incl. the NaN policy (every key containing the NaN channel's name is NaN in those windows)
    local = pd.read_pickle(str(out) + ".local")
    ch2 = chmod.get_default_channels_from_data(data)
    ch2.loc[3, "type"] = "lfp"; ch2.loc[3, "rereference"] = "ch0"; ch2.loc[3, "new_name"] = "ch3_ch0"
    ch2.loc[4, "type"] = "lfp"; ch2.loc[4, "rereference"] = "ch3&ch1"; ch2.loc[4, "new_name"] = "ch4_ch3ch1"
    data2 = data.copy()
    data2[1, 1200:1210] = np.nan
    single2 = Stream(1000.0, channels=ch2, settings=s, line_noise=50, lib=lib).run(data2, save_csv=False)
    assert list(local.columns) == list(single2.columns)
    a2, b2 = local.to_numpy(float), single2.to_numpy(float)
    assert np.array_equal(np.isnan(a2), np.isnan(b2)) and np.isnan(a2).any()
    # the group sum reaches the kernel as ONE fp32 number per sample here and as a float64 partial sum in the
    # single-process kernel: agreement to fp32 rounding of the re-referenced samples, not bit equality
    from tests import parity
    keys = list(single2.columns)[:-1]
    from oracle import nm_oracle as orc
    st2, en2, _ = orc.window_schedule(data2.shape[1], 1000.0, s.sampling_rate_features_hz, s.segment_length_features_ms)
    pv2 = parity.PipelineVerifiers(s, ch2, 1000.0, data2, st2, 1000, line_noise=50)
    for r in range(len(a2)):
        ok = ~np.isnan(b2[r, :-1])
        if loose:
            np.testing.assert_allclose(a2[r, :-1][ok], b2[r, :-1][ok], rtol=1e-3, atol=1e-4)
            continue
        # (two fp32 paths: a sharp-wave trough whose decision margin is at rounding level may flip between them -- the
        # oracle's pre-processed window says whether a miss is one of those)
        n_bad, rep, _ = parity.compare([k for k, o in zip(keys, ok) if o], a2[r, :-1][ok], b2[r, :-1][ok], s, 1000.0, 200.0, 1000,
                                       verifier=pv2.row(r))
        assert n_bad == 0, f"row {r}\n{rep}"

    # ragged window lengths: both input forms against the one-plan stream (bursts, Kalman filters, z-score, NaN policy)
    s3 = NMSettings.get_default()
    s3.reset()
    s3.features.fft = True
    s3.features.bursts = True
    s3.features.bandpass_filter = True
    s3.bandpass_filter_settings.kalman_filter = True
    s3.kalman_filter_settings.frequency_bands = ["theta", "low_beta"]
    s3.bursts_settings.time_duration_s = 2
    s3.preprocessing = ["re_referencing"]
    s3.postprocessing.feature_normalization = True
    rng3 = np.random.default_rng(19)
    t3 = np.arange(4500) / 1111.111
    data3 = rng3.standard_normal((5, 4500)) * 20 + 15 * np.sin(2 * np.pi * 18 * t3) * (np.sin(2 * np.pi * 0.7 * t3) > 0)
    data3[2, 2000:2003] = np.nan
    os.environ["NMX_CAR_FAST"] = "0"
    try:
        single3 = Stream(1111.111, data=data3, settings=s3, line_noise=50, lib=lib).run(save_csv=False)
    finally:
        del os.environ["NMX_CAR_FAST"]
    for tag in ("ragged", "ragged_local"):
        got3 = pd.read_pickle(str(out) + "." + tag)
        assert list(got3.columns) == list(single3.columns), tag
        a3, b3 = got3.to_numpy(float), single3.to_numpy(float)
        assert a3.shape == b3.shape and len(a3) > 25, tag
        assert np.array_equal(np.isnan(a3), np.isnan(b3)) and np.isnan(a3).any(), tag
        np.testing.assert_allclose(np.nan_to_num(a3), np.nan_to_num(b3), rtol=1e-3 if loose else 2e-5, atol=1e-4 if loose else 2e-6,
                                   err_msg=tag)

    # rails: the sharded table falls in the reference's classes entry by entry (tests/parity_cases.py: case_inf_members)
    from tests import parity_cases as pc
    for tag in ("inf", "inf_local"):
        got4 = pd.read_pickle(str(out) + "." + tag)
        pc.case_inf_members(lib, run=lambda: got4)


def test_multi_device_stream_runs_user_registered_features():
    """Plugins registered with add_custom_feature see the window over ALL channels (features/feature_processor.py:52-53):
    the parts of a multi-device stream hand back the pre-processed windows of their channel blocks, the coordinator
    joins them -- same table as the one-plan stream, plugin columns, z-score and NaN policy included; one rank per
    GPU (ShardedStream) cannot call them and says so."""
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge

    import py_neuromodulation_amd as nmx
    from py_neuromodulation_amd import NMSettings, _lib
    from py_neuromodulation_amd import channels as chmod
    from py_neuromodulation_amd.sharding import ShardedStream
    from py_neuromodulation_amd.stream import Stream
    from tests import user_plugins as up

    lib = _lib.NmxLibrary(ge.build_emu())
    rng = np.random.default_rng(5)
    data = rng.standard_normal((5, 2300)) * 20 + rng.uniform(-100, 100, (5, 1))
    data[3, 1700:1704] = np.nan
    nmx.add_custom_feature("channel_mean", up.ChannelMean)
    nmx.add_custom_feature("hop_stats", up.HopStats)
    os.environ["NMX_CAR_FAST"] = "0"
    try:
        s = NMSettings.get_default()
        s.preprocessing = ["notch_filter", "re_referencing"]
        one_st = Stream(1000.0, data=data, settings=s, line_noise=50, lib=lib)
        one = one_st.run(save_csv=False)
        st = Stream(1000.0, data=data, settings=s, line_noise=50, lib=lib, devices=[0, 0, 0])
        many = st.run(save_csv=False)
        assert list(many.columns) == list(one.columns)
        assert [c for c in one.columns if c.startswith("channel_mean_")] == [f"channel_mean_{n}" for n in one_st.data_processor.ch_names_used]
        a, b = many.to_numpy(float), one.to_numpy(float)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a).any()
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6, equal_nan=True)
        # window by window (DataProcessor.process, the reference's call shape) == the batch, stateful plugin included
        dp = Stream(1000.0, data=data, settings=s, line_noise=50, lib=lib).data_processor
        for r in range(3):
            row = dp.process(data[:, r * 100:r * 100 + 1000])
            assert list(row) == list(one.columns)[:-1]
            np.testing.assert_allclose(np.array(list(row.values())), b[r, :-1], rtol=1e-6, atol=1e-6, equal_nan=True)
        with pytest.raises(NotImplementedError, match="user-registered features"):
            ShardedStream(1000.0, chmod.get_default_channels_from_data(data), s, rank=0, world_size=2, lib=lib).run(data)
    finally:
        del os.environ["NMX_CAR_FAST"]
        nmx.remove_custom_feature("channel_mean")
        nmx.remove_custom_feature("hop_stats")
    # nothing registered any more: the plain table again
    plain = Stream(1000.0, data=data, settings=NMSettings.get_default(), line_noise=50, lib=lib).run(save_csv=False)
    assert not [c for c in plain.columns if "channel_mean" in c or "hops_seen" in c]


def test_multi_device_parts_on_one_device_do_not_share_staging():
    """two parts that name the same device shared one page-locked staging pool and wrote it from two host
    threads.  Every part takes its own pool slot; a batch beyond the 2^18-cell staging threshold must equal the
    one-plan stream."""
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge

    from py_neuromodulation_amd import NMSettings, _lib
    from py_neuromodulation_amd.stream import Stream

    lib = _lib.NmxLibrary(ge.build_emu())
    s = NMSettings.get_default()
    s.features.disable_all()
    for f in ("fft", "welch", "stft", "raw_hjorth", "linelength"):
        setattr(s.features, f, True)
    s.preprocessing = []
    s.postprocessing.feature_normalization = False
    s.sampling_rate_features_hz = 100
    rng = np.random.default_rng(3)
    data = rng.standard_normal((6, 1000 + 10 * 6000)) * 10
    one = Stream(1000.0, data=data, settings=s, lib=lib).run(save_csv=False).to_numpy(float)
    st = Stream(1000.0, data=data, settings=s, lib=lib, devices=[0, 0])
    assert st.data_processor.parts[0].engine._pinned is not st.data_processor.parts[1].engine._pinned
    many = st.run(save_csv=False).to_numpy(float)
    assert one.shape[0] * (one.shape[1] - 1) // 2 >= 1 << 18   # each part's output crosses the staging threshold
    np.testing.assert_array_equal(many, one)


@pytest.mark.gpu
def test_multi_device_stream_with_ragged_window_lengths_on_the_gpu():
    """The same on the product library (two plans per window length on the box's GPU(s)): fp32 kernels pair other
    channels in a shard than in the one-plan stream, and the z-score divides by small spreads."""
    from py_neuromodulation_amd import _lib

    lib = _lib.get_library()
    _ragged_multi_device_case(lib, [0, 1] if lib.device_count() >= 2 else [0, 0], rtol=2e-3, atol=2e-3)


def test_multi_device_stream_with_ragged_window_lengths():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge

    from py_neuromodulation_amd import _lib

    _ragged_multi_device_case(_lib.NmxLibrary(ge.build_emu()), [0, 0, 0], rtol=2e-5, atol=2e-6)


def _ragged_multi_device_case(lib, devices, rtol, atol):
    """A sampling rate that is not an integer (1111.111 Hz: windows of 1111 and 1112 samples) on several devices: one
    set of plans per window length, the burst histories / Kalman filters of every device travel between them where the
    length changes, the feature normaliser runs per part over all hops, a registered plugin sees the joined windows in
    hop order.  Same table as the one-plan stream (which the reference golden ragged_bursts.npz pins)."""
    sys.path.insert(0, str(ROOT))
    import py_neuromodulation_amd as nmx
    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.stream import Stream
    from tests import user_plugins as up

    rng = np.random.default_rng(41)
    sfreq, T = 1111.111, 6000
    t = np.arange(T) / sfreq
    data = rng.standard_normal((5, T)) * 20 + 15 * np.sin(2 * np.pi * 18 * t) * (np.sin(2 * np.pi * 0.7 * t) > 0)
    data[2, 3000:3003] = np.nan
    s = NMSettings.get_default()
    s.reset()
    s.features.fft = True
    s.features.bursts = True
    s.features.bandpass_filter = True
    s.bandpass_filter_settings.kalman_filter = True
    s.kalman_filter_settings.frequency_bands = ["theta", "low_beta"]
    s.bursts_settings.time_duration_s = 2
    s.preprocessing = ["re_referencing"]
    s.postprocessing.feature_normalization = True
    for plugins in (False, True):
        if plugins:
            nmx.add_custom_feature("hop_stats", up.HopStats)
        os.environ["NMX_CAR_FAST"] = "0"   # (the one-plan stream through the same row form as the parts: equal to rounding)
        try:
            one = Stream(sfreq, data=data, settings=s, line_noise=50, lib=lib).run(save_csv=False)
            many = Stream(sfreq, data=data, settings=s, line_noise=50, lib=lib, devices=devices).run(save_csv=False)
        finally:
            os.environ.pop("NMX_CAR_FAST", None)
            if plugins:
                nmx.remove_custom_feature("hop_stats")
        assert list(many.columns) == list(one.columns)
        a, b = many.to_numpy(float), one.to_numpy(float)
        assert a.shape == b.shape and len(a) > 40
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a).any()
        np.testing.assert_allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=rtol, atol=atol)

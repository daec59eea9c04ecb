"""Build-container-only check (skipped wherever /root/reference is absent, i.e. on the GPU box): the engine's
plugin classes swapped into the UNMODIFIED reference make the reference's own
``nm.Stream.run`` reproduce its own golden DataFrame.  Runs tests/golden/check_dropin_swap.py in a child
process (the swap patches module attributes of the imported reference)."""

import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(not Path("/root/reference/py_neuromodulation").is_dir(),
                    reason="the reference is only present in the build container")
def test_reference_stream_with_swapped_plugins_reproduces_its_golden():
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "check_dropin_swap.py")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("0 entries outside the parity policy") == 3, r.stdout


def _reference_suite(*flags):
    import re

    r = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "run_reference_tests.py"), *flags],
                       capture_output=True, text=True, timeout=1800, cwd="/tmp")
    failed = sorted(set(re.findall(r"^FAILED \S*?tests/(test_\S+)", r.stdout, re.M)))
    m = re.search(r"(?:(\d+) failed, )?(\d+) passed", r.stdout)
    assert m, r.stdout[-3000:] + r.stderr[-3000:]
    return failed, int(m.group(2))


@pytest.mark.skipif(not Path("/root/reference/tests").is_dir(), reason="the reference is only present in the build container")
def test_reference_own_tests_pass_with_swapped_plugins():
    """The reference's OWN in-scope test files (18 files, 87 test cases; read in place) against the reference with the nine
    feature classes, the pre-processors and the two filter classes swapped for the engine's: everything the unmodified
    reference passes under the same shim passes swapped.  (What fails in both: test_all_features.py, which enables the
    out-of-scope features whose third-party packages this image lacks.)"""
    plain_failed, plain_passed = _reference_suite("--plain")
    swap_failed, swap_passed = _reference_suite()
    assert set(swap_failed) <= set(plain_failed), (swap_failed, plain_failed)
    assert swap_passed >= plain_passed >= 84, (swap_passed, plain_passed)
    assert all(f.startswith("test_all_features.py") for f in swap_failed), swap_failed
    # ... and with `nm.Stream` itself replaced by the engine's fused Stream (the reference's pydantic settings object in,
    # the reference's DataFrame / files out)
    stream_failed, stream_passed = _reference_suite("--stream")
    assert set(stream_failed) <= set(plain_failed) and stream_passed >= plain_passed, (stream_failed, stream_passed)
    # ... and with the reference's own Stream loop over the engine's DataProcessor (`.process(window)` once per hop, windows
    # of two lengths at 1111.111 Hz, `.save_sidecar / _settings / _channels` after the loop)
    proc_failed, proc_passed = _reference_suite("--processor")
    assert set(proc_failed) <= set(plain_failed) and proc_passed >= plain_passed, (proc_failed, proc_passed)


@pytest.mark.skipif(not Path("/root/reference/examples").is_dir(), reason="the reference is only present in the build container")
def test_reference_examples_run_in_place_with_the_fused_stream():
    """The reference's example scripts that need nothing this image lacks -- examples/plot_6_real_time_demo.py (the
    one-window call, `stream.data_processor.process(window)`) and examples/plot_2_example_add_feature.py (a user feature
    registered through `nm.add_custom_feature`, then `Stream.run`) -- executed in place with `nm.Stream` = the engine's."""
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "run_reference_tests.py"), "--stream", "--examples"],
                       capture_output=True, text=True, timeout=900, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for name in ("plot_6_real_time_demo.py", "plot_2_example_add_feature.py"):
        assert f"example {name}: ran to its end, Stream = py_neuromodulation_amd.stream.Stream" in r.stdout, r.stdout[-2000:]
    assert "plot_2_example_add_feature.py:feature_df" in r.stdout
    # the reference's own FeatureReader loads what the fused Stream wrote (FEATURES.csv, SIDECAR.json, SETTINGS.yaml, channels.csv)
    assert "FeatureReader on the files of py_neuromodulation_amd.stream.Stream.run: table (51, 17) identical = True" in r.stdout
    # the same scripts with the reference's own Stream over the engine's DataProcessor
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "run_reference_tests.py"), "--processor", "--examples"],
                       capture_output=True, text=True, timeout=900, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": ran to its end, Stream = py_neuromodulation.stream.stream.Stream") == 2, r.stdout[-2000:]
    assert "identical = True" in r.stdout

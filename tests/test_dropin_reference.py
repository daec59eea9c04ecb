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

"""pytest configuration: registers the ``gpu`` marker, puts the repo root on sys.path, skips the
``gpu`` tests when no HIP device is visible, and prints how many tolerance misses each test accepted
on the strength of a per-entry conditioning report (tests/parity.py) -- zero unverified forgiveness."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

_FORGIVEN: list[tuple[str, int, dict, list]] = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        from py_neuromodulation_amd import _lib

        return _lib.get_library().device_count() >= 1
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if gpu_items and not _have_gpu():
        skip = pytest.mark.skip(reason="no HIP device / libnmx.so (the GPU parity tests run on the MI355X box)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _scratch_cwd(tmp_path, monkeypatch):
    """Stream.run always leaves {name}_SIDECAR.json / _SETTINGS.yaml / _channels.csv under out_dir (default:
    the working directory), like the reference (stream/stream.py:338): keep them out of the repository."""
    monkeypatch.chdir(tmp_path)


@pytest.fixture(autouse=True)
def _parity_stats(request):
    from tests import parity

    parity.reset_stats()
    yield
    st = parity.STATS
    if st["compared"]:
        _FORGIVEN.append((request.node.nodeid, st["compared"], dict(st["forgiven"]), list(st["notes"])))


def pytest_terminal_summary(terminalreporter):
    if not _FORGIVEN:
        return
    tr = terminalreporter
    tr.write_sep("-", "parity: entries compared / misses accepted on a verified conditioning report")
    for nodeid, n, forgiven, notes in _FORGIVEN:
        tot = sum(forgiven.values())
        tr.write_line(f"{nodeid}: {n} compared, {tot} accepted {forgiven if tot else ''}")
        for note in notes[:6]:
            tr.write_line(f"    {note}")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"

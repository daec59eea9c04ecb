"""Import shim that makes /root/reference importable in the BUILD container only.

Used by tests/golden/make_golden.py to generate the golden vectors; it is never
executed on the GPU box (there is no /root/reference there) and nothing in
``py_neuromodulation_amd`` or the tests imports it.

What it does:
  1. ``importlib.metadata.version("py_neuromodulation")`` -> "0.1.4"
     (py_neuromodulation/__init__.py:12 would raise PackageNotFoundError).
  2. stubs the third-party packages that are not installed in this image
     (mne, mne_lsl, mne_bids, ... ) with permissive dummy modules.
  3. injects oracle.mne_restated.{create_filter,_overlap_add_filter,resample} as
     ``mne.filter.*`` so the reference's OWN MNEFilter / BandPower / Bursts /
     SharpwaveAnalyzer / NotchFilter code runs unmodified on top of the restated taps
     (tap design itself stays "parity unpinned", see oracle/__init__.py).
"""

from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.metadata as _md
import sys
import types
from pathlib import Path

REFERENCE_ROOT = "/root/reference"
_STUBBED = {
    "mne", "mne_lsl", "mne_bids", "mne_connectivity", "fooof", "nolds", "pybispectra",
    "numba", "cbor2", "webview", "nibabel", "seaborn", "skopt", "imblearn", "pyparrm",
    "skops", "mrmr", "llvmlite",
}


class _DummyMeta(type):
    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return cls


class _Dummy(metaclass=_DummyMeta):
    def __init__(self, *a, **k):
        raise ImportError("stubbed third-party symbol used at run time")


class _StubModule(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def load_reference():
    """Return the imported reference package (py_neuromodulation)."""
    if not Path(REFERENCE_ROOT).is_dir():
        raise RuntimeError("/root/reference is only present in the build container")
    orig_version = _md.version
    _md.version = lambda n: "0.1.4" if n == "py_neuromodulation" else orig_version(n)
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    repo = str(Path(__file__).resolve().parents[2])
    if repo not in sys.path:
        sys.path.insert(0, repo)
    import py_neuromodulation as nm  # noqa: E402

    nm.logger.set_level("ERROR")
    import mne.filter as mf  # the stub

    from oracle import mne_restated

    mf.create_filter = mne_restated.create_filter
    mf._overlap_add_filter = mne_restated._overlap_add_filter
    mf.resample = mne_restated.resample
    return nm

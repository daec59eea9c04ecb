"""Shared helpers for the parity tests (golden loading, settings reconstruction)."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_golden(name: str):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


def settings_from_json(js) -> "NMSettings":
    """Rebuild the engine's NMSettings from the reference's ``model_dump()`` JSON."""
    from py_neuromodulation_amd.settings import NMSettings

    return NMSettings(**json.loads(str(js)))


def golden_dict(g, prefix: str) -> dict:
    return dict(zip([str(k) for k in g[prefix + "_keys"]], g[prefix + "_values"]))


def assert_dict_close(got: dict, want: dict, rtol: float, atol: float = 0.0, what: str = ""):
    assert list(got.keys()) == list(want.keys()), f"{what}: key order/name mismatch"
    a = np.array([float(v) for v in got.values()])
    b = np.array([float(v) for v in want.values()])
    bad = ~np.isclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    if bad.any():
        keys = np.array(list(want.keys()))[bad]
        msg = "\n".join(f"  {k}: got {x!r} want {y!r}" for k, x, y in
                        list(zip(keys, a[bad], b[bad]))[:12])
        raise AssertionError(f"{what}: {bad.sum()} / {len(b)} values differ\n{msg}")

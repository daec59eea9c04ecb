#!/usr/bin/env python
"""Sweep of the two randomised-settings generators (tests/parity_cases.py) over many seeds on the GPU library; prints
one line per failing seed.  Not collected by pytest (test infrastructure, imports the oracle):
    python tests/fuzz_sweep.py 1000 1300          # seeds [1000, 1300) of the five generators"""
import os
import sys
import time
import warnings
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
warnings.filterwarnings("ignore")


def main(lo, hi, budget_s=1500.0):
    from py_neuromodulation_amd import _lib
    from tests import parity_cases as pc

    if os.environ.get("NMX_FUZZ_EMU"):   # the CPU logic emulator of tests/test_kernel_logic_emu.py instead of the GPU library
        import __graft_entry__ as ge
        lib = _lib.NmxLibrary(ge.build_emu())
    else:
        lib = _lib.get_library()
    import tempfile

    os.chdir(tempfile.mkdtemp(prefix="nmx_fuzz_"))   # Stream.run leaves its side-car files under the working directory
    only = os.environ.get("NMX_FUZZ_ONLY")   # "narrow" / "wide" / "channels" / "bursts" / "windows" / "highrate"
    t0 = time.time()
    n = bad = 0
    for seed in range(lo, hi):
        for name, fn in (("narrow", pc.case_random_settings), ("wide", pc.case_random_settings_wide),
                         ("channels", pc.case_random_channel_tables), ("bursts", pc.case_random_burst_streams),
                         ("windows", pc.case_random_window_by_window), ("highrate", pc.case_random_settings_highrate),
                         ("windows_wide", lambda lib, seed: pc.case_random_window_by_window(lib, seed, wide=True))):
            if only and name != only:
                continue
            if time.time() - t0 > budget_s:
                print(f"time budget reached at seed {seed}: {n} cases, {bad} failures")
                return
            n += 1
            t1 = time.time()
            try:
                fn(lib, seed)
            except BaseException as e:   # noqa: BLE001
                bad += 1
                print(f"FAIL {name} {seed} [{time.time() - t1:.1f}s] {type(e).__name__}: {str(e)[:int(os.environ.get('NMX_FUZZ_MSG', '700'))]}", flush=True)
    print(f"{n} cases, {bad} failures, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0)

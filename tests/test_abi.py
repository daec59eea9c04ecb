"""The C-ABI boundary: libnmx.so builds, loads and exports every function include/nmx.h declares.

No compute call is made here (there is no GPU in the CPU test tier); the entry points that must work
without a device (version, device count, error string, argument validation) are exercised.
"""

from __future__ import annotations

import ctypes as C
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def declared_functions() -> list[str]:
    text = (ROOT / "include" / "nmx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(nmx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    path = g.build_lib()
    return C.CDLL(str(path))


def test_header_declares_the_documented_entry_points():
    names = declared_functions()
    for must in ("nmx_plan_create", "nmx_process_batch", "nmx_process_window", "nmx_preprocess_window",
                 "nmx_filter_window", "nmx_state_export", "nmx_norm_process", "nmx_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"libnmx.so lacks {missing}"


def test_library_exports_nothing_but_the_declared_symbols():
    """The converse: built with -fvisibility=hidden and a version script generated from the header, the dynamic symbol
    table holds the header's functions and nothing else -- the launchers the translation units call each other through
    (nmx_w64c_launch_rd64, nmx_specmm_launch, nmxi_note_kernel ...), the kernels' host-side handles and libstdc++'s
    template instantiations are not an API anybody was promised."""
    import subprocess

    import __graft_entry__ as g

    out = subprocess.run(["nm", "-D", "--defined-only", str(g.build_lib())], check=True, capture_output=True, text=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert names == declared_functions()


def test_python_binding_lists_every_declared_symbol():
    from py_neuromodulation_amd import _lib

    assert sorted(_lib._EXPORTS) == declared_functions()


def test_entry_points_that_need_no_device(lib):
    lib.nmx_abi_version.restype = C.c_int
    lib.nmx_device_count.restype = C.c_int
    lib.nmx_last_error.restype = C.c_char_p
    from py_neuromodulation_amd._lib import NMX_ABI_VERSION

    assert lib.nmx_abi_version() == NMX_ABI_VERSION
    assert lib.nmx_device_count() >= 0
    # argument validation happens before any device work: error code + thread-local message
    lib.nmx_plan_create.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.nmx_plan_create(None, None) < 0
    assert lib.nmx_last_error()
    lib.nmx_norm_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
    h = C.c_void_p()
    assert lib.nmx_norm_create(0, 0, 1, 3.0, 300, None, C.byref(h)) < 0    # n_cols must be positive
    assert b"n_cols" in lib.nmx_last_error()


def test_product_loader_has_no_cpu_fallback(tmp_path):
    from py_neuromodulation_amd._lib import NmxError, NmxLibrary

    with pytest.raises(NmxError):
        NmxLibrary(tmp_path / "missing_libnmx.so")


def test_product_package_has_no_host_compute_path():
    """The package neither imports the oracle nor scikit-learn / scipy.signal for a compute fall-back: every
    normalisation method is a device kernel."""
    import re
    from pathlib import Path

    from py_neuromodulation_amd import NMSettings
    from py_neuromodulation_amd.processing import DeviceFeatureNormalizer, FeatureNormalizer

    pkg = Path(__file__).resolve().parents[1] / "py_neuromodulation_amd"
    for f in pkg.glob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(import|from)\s+(sklearn|oracle)", src, re.M), f.name
    # all eight methods of the reference (processing/normalization.py:57-70) are device kernels
    assert set(DeviceFeatureNormalizer.METHODS) == {"mean", "median", "zscore", "zscore-median", "robust", "minmax",
                                                    "quantile", "power"}
    s = NMSettings.get_default()
    s.feature_normalization_settings.normalization_method = "no-such-method"
    with pytest.raises(NotImplementedError, match="no-such-method"):
        FeatureNormalizer(s)


def test_abi_from_plain_c(tmp_path):
    """include/nmx.h is valid C and the library links and runs from a C program."""
    import subprocess

    import __graft_entry__ as g

    lib = g.build_lib()
    exe = tmp_path / "abi_smoke"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", str(ROOT / "tests" / "c_abi" / "abi_smoke.c"), "-I", str(ROOT / "include"),
           "-L", str(lib.parent), "-lnmx", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_bench_refuses_missing_gpus():
    """`python bench.py --gpus N` without a launcher must start N ranks itself or fail loudly when fewer than N devices
    are visible -- never a 1-GPU number under an N-GPU label; under a launcher --gpus must equal WORLD_SIZE."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NMX_BENCH_FORCE_DEVICE")}
    try:
        import torch

        n_dev = torch.cuda.device_count()
    except Exception:
        n_dev = 0
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(n_dev + 7)], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "HIP device(s) visible" in (r.stdout + r.stderr)
    assert '"n_gpus"' not in r.stdout
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "!= WORLD_SIZE" in (r.stdout + r.stderr)


def test_host_staging_helpers_equal_numpy():
    """nmx_host_stage_rows / nmx_host_group_sums / nmx_host_widen_rows (host passes, no device): bit for bit what the
    NumPy expressions they replace give -- cast after the float64 subtraction, nan_to_num before the float64 sum in row
    order, widening into column runs -- on strided views, row subsets, and sizes on both sides of the threading cut."""
    import numpy as np

    from py_neuromodulation_amd import _lib
    from py_neuromodulation_amd.engine import parallel_cast

    lib = _lib.NmxLibrary()
    rng = np.random.default_rng(5)
    for (R, T) in ((3, 50), (9, 4000), (40, 30011)):
        big = rng.standard_normal((R + 2, T + 7)) * 1e3 + rng.uniform(-1e6, 1e6, (R + 2, 1))
        big[1, 3] = np.nan
        big[2, 5] = np.inf
        big[0, 6] = -np.inf
        for dtype in (np.float64, np.float32):
            src = big.astype(dtype)[1:R + 1, 2:T + 2]           # a strided view
            sub = rng.uniform(-1e6, 1e6, R) * (rng.random(R) < 0.7)
            want = (np.asarray(src, np.float64) - sub[:, None]).astype(np.float32)
            got = np.full((R, T + 5), -7.0, np.float32)
            parallel_cast(got[:, :T], src, sub, lib)
            np.testing.assert_array_equal(got[:, :T], want)
            assert (got[:, T:] == -7.0).all()
            got2 = np.empty((R, T), np.float32)
            parallel_cast(got2, src, None, lib)
            np.testing.assert_array_equal(got2, src.astype(np.float32))
            # a row subset, a column range
            rows = np.ascontiguousarray(rng.permutation(R)[:max(1, R // 2)], dtype=np.int32)
            t0, t1 = T // 5, T - 3
            dst = np.zeros((len(rows), T), np.float32)
            lib.check(lib.lib.nmx_host_stage_rows(dst.ctypes.data, T, src.ctypes.data, int(dtype == np.float64),
                                                  src.strides[0] // src.itemsize, rows.ctypes.data, len(rows), t0, t1,
                                                  sub.ctypes.data, 3))
            np.testing.assert_array_equal(dst[:, t0:t1], want[rows][:, t0:t1])
            assert (dst[:, :t0] == 0).all() and (dst[:, t1:] == 0).all()
            # group sums: nan_to_num of the float32-rounded samples, float64, rows in order
            s = np.full(T, -1.0)
            lib.check(lib.lib.nmx_host_group_sums(s.ctypes.data, src.ctypes.data, int(dtype == np.float64),
                                                  src.strides[0] // src.itemsize, rows.ctypes.data, len(rows), t0, t1, 0))
            ref = np.zeros(T)
            for r in rows:
                ref += np.nan_to_num(src[r].astype(np.float32)).astype(np.float64)
            np.testing.assert_array_equal(s[t0:t1], ref[t0:t1])
            assert (s[:t0] == -1.0).all() and (s[t1:] == -1.0).all()
        # rows out and group sums in one pass (nmx_host_stage_parts): two parts, a row nobody takes, two groups
        for dtype in (np.float64, np.float32):
            src = big.astype(dtype)[1:R + 1, 2:T + 2]
            parts = [np.full((R, T), -5.0, np.float32), np.full((R, T), -5.0, np.float32)]
            dst = np.zeros(R, np.uint64)
            owner = {}
            for j in range(R - 1):           # (the last row goes nowhere)
                k = j % 2
                owner[j] = (k, j // 2)
                dst[j] = parts[k].ctypes.data + (j // 2) * parts[k].strides[0]
            groups = [np.arange(0, R, 2, dtype=np.int32), np.arange(R, dtype=np.int32)[::-1].copy()]
            gptr = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int32)
            grows = np.concatenate(groups).astype(np.int32)
            sums = [np.full(T, -1.0) for _ in groups]
            sptr = np.array([v.ctypes.data for v in sums], np.uint64)
            t0, t1 = 3, T - 2
            lib.check(lib.lib.nmx_host_stage_parts(src.ctypes.data, int(dtype == np.float64), src.strides[0] // src.itemsize, R,
                                                   t0, t1, dst.ctypes.data, 2, gptr.ctypes.data, grows.ctypes.data,
                                                   sptr.ctypes.data, 0))
            for j, (k, r) in owner.items():
                np.testing.assert_array_equal(parts[k][r, t0:t1], src[j, t0:t1].astype(np.float32))
                assert (parts[k][r, :t0] == -5.0).all() and (parts[k][r, t1:] == -5.0).all()
            for g, v in zip(groups, sums):
                ref = np.zeros(T)
                for r in g:
                    ref += np.nan_to_num(src[r].astype(np.float32)).astype(np.float64)
                np.testing.assert_array_equal(v[t0:t1], ref[t0:t1])
                assert (v[:t0] == -1.0).all() and (v[t1:] == -1.0).all()
        # widening into column runs
        f = rng.standard_normal((R, T)).astype(np.float32)
        f[0, 0] = np.nan
        table = np.full((R, 2 * T + 3), -3.0)
        cut = T // 3
        runs = np.array([[T + 3, 0, cut], [1, cut, T - cut]], dtype=np.int64)
        lib.check(lib.lib.nmx_host_widen_rows(table.ctypes.data, table.shape[1], f.ctypes.data, T, 0, R, runs.ctypes.data, 2, 0))
        np.testing.assert_array_equal(table[:, T + 3:T + 3 + cut], f[:, :cut].astype(np.float64))
        np.testing.assert_array_equal(table[:, 1:1 + T - cut], f[:, cut:].astype(np.float64))
        assert (table[:, 0] == -3.0).all() and (table[:, 1 + T - cut:T + 3] == -3.0).all() and (table[:, T + 3 + cut:] == -3.0).all()
        o64 = np.empty((R, T))
        parallel_cast(o64, f, None, lib)
        np.testing.assert_array_equal(o64, f.astype(np.float64))
    assert lib.lib.nmx_host_widen_rows(None, 0, None, 0, 0, 0, None, 0, 0) != 0   # argument validation, no crash


def test_host_staging_helpers_from_several_threads_and_after_fork():
    """The helpers' worker pool takes calls from several host threads at once (a multi-device stream stages on one thread
    and widens on one per plan), and a forked child gets workers of its own."""
    import os
    import threading

    import numpy as np

    from py_neuromodulation_amd import _lib
    from py_neuromodulation_amd.engine import parallel_cast

    lib = _lib.NmxLibrary()
    rng = np.random.default_rng(9)
    srcs = [rng.standard_normal((40 + k, 9000)) for k in range(4)]
    bad: list = []

    def job(k):
        try:
            for _ in range(40):
                dst = np.empty(srcs[k].shape, np.float32)
                parallel_cast(dst, srcs[k], None, lib)
                if not np.array_equal(dst, srcs[k].astype(np.float32)):
                    bad.append(k)
                o = np.empty(dst.shape)
                parallel_cast(o, dst, None, lib)
                if not np.array_equal(o, dst.astype(np.float64)):
                    bad.append(-k - 1)
        except BaseException as e:   # noqa: BLE001
            bad.append(repr(e))

    th = [threading.Thread(target=job, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad
    pid = os.fork()
    if pid == 0:   # the child: the parent's workers do not exist here
        try:
            dst = np.empty(srcs[0].shape, np.float32)
            parallel_cast(dst, srcs[0], None, lib)
            os._exit(0 if np.array_equal(dst, srcs[0].astype(np.float32)) else 3)
        except BaseException:   # noqa: BLE001
            os._exit(4)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_result_tables_are_recycled_but_never_shared():
    """engine.table_empty: a dropped table's memory serves the next one of its size; a table still referenced (directly,
    through a view, or inside a DataFrame) is never handed out again; small tables bypass the pool."""
    import gc

    import numpy as np
    import pandas as pd

    from py_neuromodulation_amd.engine import _TABLES, release_tables, table_empty

    release_tables()
    shape = (700, 1000)   # 5.6 MB: pooled
    a = table_empty(shape)
    a[:] = 1.0
    addr = a.ctypes.data
    b = table_empty(shape)
    assert b.ctypes.data != addr and not np.shares_memory(a, b)
    view = a[10:20]
    df = pd.DataFrame(a, columns=[str(i) for i in range(shape[1])])
    del a
    gc.collect()
    c = table_empty(shape)
    assert c.ctypes.data != addr          # the view and the frame still hold it
    assert float(view[0, 0]) == 1.0 and float(df.iloc[5, 5]) == 1.0
    del view, df
    gc.collect()
    d = table_empty(shape, np.nan)
    assert d.ctypes.data == addr and np.isnan(d).all() and d.flags.writeable and d.flags.c_contiguous
    small = table_empty((4, 4))
    assert small.base is None or not isinstance(small.base, type(d.base))
    del b, c, d
    gc.collect()
    assert len(_TABLES._free) <= _TABLES.keep
    assert release_tables() >= 1 and len(_TABLES._free) == 0

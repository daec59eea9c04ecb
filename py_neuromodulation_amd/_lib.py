"""ctypes binding of libnmx.so (C ABI in include/nmx.h).

The product path has NO CPU fallback: if libnmx.so is missing, or no HIP device is visible
when a plan is created, an exception is raised.  (``NmxLibrary(path=...)`` exists so the
CPU-only test-suite can point the same binding at its test-only logic emulator; nothing in
this package passes ``path``.)
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

NMX_ABI_VERSION = 11
NMX_MAX_BANDS = 16
NMX_MAX_FILTERS = 24
NMX_MAX_SW_COMBOS = 48

# feature bits (FeatureSelector order, stream/settings.py:41-55)
F_HJORTH, F_RAW, F_BANDPOWER, F_STFT, F_FFT, F_WELCH, F_SHARPWAVE, F_BURSTS, F_LINELENGTH = (
    1 << i for i in range(9))
EST_BITS = {"mean": 1, "median": 2, "std": 4, "max": 8}
SW_FEATURES = ["peak_left", "peak_right", "num_peaks", "trough", "width", "prominence",
               "interval", "decay_time", "rise_time", "sharpness", "rise_steepness",
               "decay_steepness", "slope_ratio"]
SW_ESTIMATORS = ["mean", "median", "max", "min", "var"]


class Cols(C.Structure):
    _fields_ = [("base", C.c_int32), ("ch_stride", C.c_int32), ("a_stride", C.c_int32),
                ("b_stride", C.c_int32)]


class OscDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("log_transform", C.c_int32), ("estimators", C.c_uint32),
                ("return_spectrum", C.c_int32), ("bin_lo", C.c_int32 * NMX_MAX_BANDS),
                ("bin_hi", C.c_int32 * NMX_MAX_BANDS), ("cols", Cols), ("psd_cols", Cols)]


class FilterDesc(C.Structure):
    _fields_ = [("taps", C.POINTER(C.c_double)), ("n_taps", C.c_int32), ("bp_seglen", C.c_int32),
                ("bp_band_index", C.c_int32), ("burst_index", C.c_int32), ("sw_index", C.c_int32)]


class PlanDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_channels", C.c_int32),
        ("window", C.c_int32), ("sfreq", C.c_double), ("feat_hz", C.c_double),
        ("features", C.c_uint32), ("n_outputs", C.c_int32), ("n_bands", C.c_int32),
        ("hjorth_cols", Cols), ("raw_cols", Cols), ("linelength_cols", Cols),
        ("fft", OscDesc), ("welch", OscDesc), ("stft", OscDesc),
        ("n_filters", C.c_int32), ("filters", FilterDesc * NMX_MAX_FILTERS),
        ("bp_features", C.c_uint32), ("bp_log_transform", C.c_int32), ("bp_cols", Cols),
        ("n_burst_bands", C.c_int32), ("burst_threshold", C.c_double),
        ("burst_time_duration_s", C.c_double), ("burst_out_mask", C.c_uint32),
        ("burst_cols", Cols),
        ("n_sw_filters", C.c_int32), ("sw_n_combos", C.c_int32),
        ("sw_combo_feature", C.c_int32 * NMX_MAX_SW_COMBOS),
        ("sw_combo_estimator", C.c_int32 * NMX_MAX_SW_COMBOS),
        ("sw_distance_peaks", C.c_double), ("sw_distance_troughs", C.c_double),
        ("sw_estimate_peaks", C.c_int32), ("sw_estimate_troughs", C.c_int32),
        ("sw_between", C.c_int32), ("sw_cols", Cols), ("sw_numpeaks_cols", Cols),
        ("notch_taps", C.POINTER(C.c_double)), ("n_notch_taps", C.c_int32),
        ("ref_matrix", C.POINTER(C.c_double)), ("n_channels_in", C.c_int32),
        ("bp_kalman_mask", C.c_uint32),
        ("kalman_Tp", C.c_double), ("kalman_sigma_w", C.c_double), ("kalman_sigma_v", C.c_double),
        ("raw_window", C.c_int32), ("resample_ratio", C.c_double),
        ("n_pre_filters", C.c_int32), ("pre_taps", C.POINTER(C.c_double) * 4), ("n_pre_taps", C.c_int32 * 4),
        ("raw_norm_method", C.c_int32), ("raw_norm_n", C.c_int32), ("raw_norm_add", C.c_int32),
        ("raw_norm_clip", C.c_float),
        ("segment_length_s", C.c_double),
    ]


class NmxError(RuntimeError):
    pass


_EXPORTS = [
    "nmx_abi_version", "nmx_device_count", "nmx_last_error", "nmx_plan_create",
    "nmx_plan_destroy", "nmx_plan_n_outputs", "nmx_process_batch", "nmx_process_batch_tap", "nmx_process_window",
    "nmx_preprocess_window", "nmx_filter_window", "nmx_state_reset", "nmx_state_size",
    "nmx_state_export", "nmx_state_import", "nmx_last_timing_ms", "nmx_last_kernels",
    "nmx_norm_create", "nmx_norm_destroy", "nmx_norm_process", "nmx_norm_reset",
    "nmx_norm_state_size", "nmx_norm_state_export", "nmx_norm_state_import",
    "nmx_plan_attach_norm", "nmx_host_alloc", "nmx_host_free", "nmx_device_pool_trim", "nmx_reref_f64", "nmx_resample_f64",
    "nmx_plan_carries_offsets", "nmx_plan_set_offsets", "nmx_plan_get_offsets", "nmx_plan_set_pipeline",
    "nmx_host_stage_rows", "nmx_host_group_sums", "nmx_host_stage_parts", "nmx_host_widen_rows",
]


def default_library_path() -> Path:
    return Path(__file__).resolve().parent / "libnmx.so"


class NmxLibrary:
    """Loaded libnmx.so with typed entry points."""

    def __init__(self, path: str | os.PathLike | None = None) -> None:
        self.path = Path(path) if path is not None else default_library_path()
        if not self.path.exists():
            raise NmxError(
                f"{self.path} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        if path is None:
            # Share ONE HIP runtime with torch when torch is (or will be) used in the same process: torch ships its own
            # libamdhip64.so under the same SONAME, and device pointers only mean something inside one runtime.  Importing
            # torch for that costs a second; loading its runtime library does the same job in tens of milliseconds --
            # libnmx's DT_NEEDED entry then resolves to the copy already in the process, and so does a later `import torch`.
            import sys

            if "torch" not in sys.modules:
                try:
                    import importlib.util

                    spec = importlib.util.find_spec("torch")
                    hip = Path(spec.origin).parent / "lib" / "libamdhip64.so" if spec and spec.origin else None
                    if hip is not None and hip.exists():
                        C.CDLL(str(hip), mode=C.RTLD_GLOBAL)
                except Exception:  # pragma: no cover - torch is optional plumbing
                    pass
        self.lib = C.CDLL(str(self.path))
        L = self.lib
        for name in _EXPORTS:
            if not hasattr(L, name):
                raise NmxError(f"{self.path} does not export {name}")
        L.nmx_abi_version.restype = C.c_int
        L.nmx_device_count.restype = C.c_int
        L.nmx_last_error.restype = C.c_char_p
        L.nmx_plan_create.argtypes = [C.POINTER(PlanDesc), C.POINTER(C.c_void_p)]
        L.nmx_plan_destroy.argtypes = [C.c_void_p]
        L.nmx_plan_n_outputs.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.nmx_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.nmx_process_batch_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.nmx_process_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.nmx_preprocess_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.nmx_filter_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.nmx_state_reset.argtypes = [C.c_void_p]
        L.nmx_state_size.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.nmx_state_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.nmx_state_import.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.nmx_last_timing_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.nmx_last_kernels.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]
        L.nmx_norm_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                      C.POINTER(C.c_void_p)]
        L.nmx_norm_destroy.argtypes = [C.c_void_p]
        L.nmx_norm_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        L.nmx_norm_reset.argtypes = [C.c_void_p]
        L.nmx_norm_state_size.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.nmx_norm_state_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.nmx_norm_state_import.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.nmx_plan_attach_norm.argtypes = [C.c_void_p, C.c_void_p]
        L.nmx_host_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
        L.nmx_host_free.argtypes = [C.c_void_p]
        L.nmx_device_pool_trim.argtypes = [C.c_int64, C.POINTER(C.c_int64)]
        L.nmx_resample_f64.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_int64,
                                       C.c_int64]
        L.nmx_reref_f64.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                    C.c_int64]
        L.nmx_plan_set_pipeline.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nmx_plan_carries_offsets.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.nmx_plan_set_offsets.argtypes = [C.c_void_p, C.c_void_p]
        L.nmx_plan_get_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.nmx_host_stage_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int32,
                                          C.c_int64, C.c_int64, C.c_void_p, C.c_int32]
        L.nmx_host_group_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int32, C.c_int64,
                                          C.c_int64, C.c_int32]
        L.nmx_host_stage_parts.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.nmx_host_widen_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                          C.c_int32, C.c_int32]
        if L.nmx_abi_version() != NMX_ABI_VERSION:
            raise NmxError("libnmx ABI version mismatch")

    def check(self, rc: int) -> None:
        if rc != 0:
            msg = self.lib.nmx_last_error()
            text = msg.decode() if msg else ""
            if rc == -1:
                raise ValueError(f"nmx: {text}")
            raise NmxError(f"nmx error {rc}: {text}")

    def device_count(self) -> int:
        return int(self.lib.nmx_device_count())


_default: NmxLibrary | None = None


def get_library() -> NmxLibrary:
    global _default
    if _default is None:
        _default = NmxLibrary()
    return _default

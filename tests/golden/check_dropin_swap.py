"""Drop-in check against the UNMODIFIED reference (build container only; /root/reference cannot travel).

INTEGRATION.md section 1 in executable form: import the reference through tests/golden/ref_shim.py, replace the
nine feature classes and the pre-processors it looks up by name (features/feature_processor.py:45-50,
processing/data_preprocessor.py:45-51) by the engine's plugins, run the reference's OWN ``nm.Stream.run`` --
its generator, its DataProcessor, its NaN policy, its FeatureNormalizer -- on the README demo data and
compare the DataFrame with the golden the untouched reference produced (tests/golden/pipeline_readme.npz).
This executes the claim that the plugins read the reference's pydantic ``NMSettings`` duck-typed.

There is no GPU in the build container, so the plugins are pointed at the test-only logic emulator
(tests/emu/libnmx_emu.so: the same kernel source behind the same C ABI); on a GPU box the same swap runs
on libnmx.so.

    python tests/golden/check_dropin_swap.py        -> prints one line per case, exit code 0 when all agree
"""

from __future__ import annotations

import json
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

FEATURE_CLASSES = ("Hjorth", "Raw", "BandPower", "STFT", "FFT", "Welch", "SharpwaveAnalyzer", "Bursts", "LineLength")
PREPROCESSOR_CLASSES = ("NotchFilter", "ReReferencer", "Resampler", "PreprocessingFilter")


def run(verbose: bool = True) -> int:
    from tests import parity   # OUR tests package first: the reference ships a `tests` package too
    from tests.helpers import load_golden, settings_from_json

    import ref_shim

    nm = ref_shim.load_reference()
    import py_neuromodulation.features as nmf
    import py_neuromodulation.processing as nmp

    import __graft_entry__ as ge
    import py_neuromodulation_amd.features as amd_f
    import py_neuromodulation_amd.processing as amd_p
    from py_neuromodulation_amd import _lib
    warnings.filterwarnings("ignore")
    if _lib._default is None:   # no GPU here: the plugins' default library := the logic emulator
        _lib._default = _lib.NmxLibrary(ge.build_emu())
    saved = {(m, c): getattr(m, c) for m, names in ((nmf, FEATURE_CLASSES), (nmp, PREPROCESSOR_CLASSES)) for c in names}
    n_fail = 0
    try:
        for c in FEATURE_CLASSES:
            setattr(nmf, c, getattr(amd_f, c))
        for c in PREPROCESSOR_CLASSES:
            setattr(nmp, c, getattr(amd_p, c))
        g = load_golden("pipeline_readme")
        data, sfreq = g["data"], float(g["sfreq"])
        for tag in ("reref_nonorm", "default_nonorm", "default"):
            s = nm.NMSettings(**json.loads(str(g[f"{tag}_settings_json"])))      # the reference's pydantic settings
            st = nm.Stream(sfreq=sfreq, data=data, settings=s, line_noise=50, verbose=False)
            used = {type(f).__module__.split(".")[0] for f in st.data_processor.features.features.values()}
            assert used == {"py_neuromodulation_amd"}, f"the reference did not pick up the swapped classes: {used}"
            usedp = {type(p).__module__.split(".")[0] for p in st.data_processor.preprocessors.preprocessors}
            assert usedp <= {"py_neuromodulation_amd"}, f"pre-processors not swapped: {usedp}"
            with tempfile.TemporaryDirectory() as td:
                df = st.run(data=data, out_dir=td, save_csv=False)
            cols = [str(c) for c in g[f"{tag}_columns"]]
            assert list(df.columns) == cols, "column order differs"
            got, want = df.to_numpy(dtype=np.float64), g[f"{tag}_values"]
            np.testing.assert_array_equal(got[:, -1], want[:, -1])
            so = settings_from_json(g[f"{tag}_settings_json"])
            bad = 0
            if so.postprocessing.feature_normalization:
                # the reference's own float64 normaliser on fp32 features: z-scores amplify by value / std
                # (tests/parity_cases.case_pipeline_readme_default_zscore verifies the stages separately)
                err = np.abs(got[1:, :-1] - want[1:, :-1])
                bad = int(np.nanmedian(err) > 1e-4) + int(np.nanmax(np.abs(got[0, :-1] - want[0, :-1]) /
                                                                  (np.abs(want[0, :-1]) + 1e-3)) > 1e-3)
            else:
                from oracle import nm_oracle as orc

                ch = json.loads(str(g[f"{tag}_channels_json"]))
                starts, ends, _ = orc.window_schedule(data.shape[1], sfreq, so.sampling_rate_features_hz,
                                                      so.segment_length_features_ms)
                pv = parity.PipelineVerifiers(so, ch, sfreq, data, starts, int(ends[0] - starts[0]), ends=ends)
                for r in range(len(got)):
                    b, rep, _ = parity.compare(cols[:-1], got[r, :-1], want[r, :-1], so, sfreq, 1.0, 1000, verifier=pv.row(r))
                    if b and verbose:
                        print(f"{tag} row {r}\n{rep}")
                    bad += b
            n_fail += bad
            if verbose:
                print(f"{tag}: reference Stream.run with swapped plugins -> {df.shape}, {bad} entries outside the parity policy")
    finally:
        for (m, c), v in saved.items():
            setattr(m, c, v)
    return n_fail


if __name__ == "__main__":
    sys.exit(1 if run() else 0)

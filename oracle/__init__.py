"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU (NumPy/SciPy, float64) restatement of py_neuromodulation's per-hop hot path
(nm.Stream -> DataProcessor.process -> filter/ -> features/).  It exists to CHECK
the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; nothing under ``py_neuromodulation_amd/`` does.

Pinning status (see DESIGN.md section "Oracle"):
  * everything computed with NumPy/SciPy in the reference (FFT/Welch/STFT band power,
    Hjorth, Raw, LineLength, FIR-bank apply, BandPower incl. Kalman smoothing, Bursts,
    SharpwaveAnalyzer, ReReferencer, PreprocessingFilter glue, window schedule, NaN policy,
    FeatureNormalizer) is PINNED: the
    goldens in ``tests/golden/`` were produced by importing the reference itself
    (``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks this
    restatement against them.
  * the three MNE-Python entry points the reference calls (``mne.filter.create_filter``,
    ``_overlap_add_filter``, ``resample``) are restated from MNE's published algorithm in
    ``oracle/mne_restated.py``; MNE is an unpinned, un-vendored dependency that is not
    installable here, so FIR *design*, notch edge handling and resampling are
    "PARITY UNPINNED" against a real MNE install.  FIR taps are an explicit input of
    every kernel and are stored in the goldens, so everything downstream of tap design
    is pinned by reference code.
"""

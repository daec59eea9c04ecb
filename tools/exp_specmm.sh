#!/bin/bash
# Mode A (FFT band power + Hjorth + LineLength, distinct windows from HBM): the matrix-pipe spectrum kernel against the
# wave-level FFT kernel on one lease.   gpurun -- 'bash tools/exp_specmm.sh'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest $R/tests/test_gpu_parity.py -x -q -k "matrix_pipe or batch_equals_window_by_window" 2>&1 | tail -15
echo "NMX_SPECMM=1"
timeout 200 python $R/tools/bench_scan.py
timeout 200 python $R/tools/bench_scan.py --features fft
echo "NMX_SPECMM=0"
NMX_SPECMM=0 timeout 200 python $R/tools/bench_scan.py

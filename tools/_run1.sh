cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix_pipe or batch_equals_window_by_window or nan or ragged" 2>&1 | tail -8
cd /tmp
for i in 1 2 3; do python $GRAFT_REPO_ROOT/tools/bench_scan.py 2>&1 | grep -v amdgpu | cut -c1-175; done
python $GRAFT_REPO_ROOT/tools/bench_scan.py --features fft 2>&1 | grep -v amdgpu | cut -c1-175

#!/bin/bash
# Round 6, after the float64 stand-alone objects: their GPU tests, timings and a kernel trace.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "standalone or plugin_classes_as or resampler or reference_property" > $O/r06_f64_tests.log 2>&1; tail -3 $O/r06_f64_tests.log
timeout 600 python tools/bench_standalone_f64.py > $O/r06_standalone_f64.jsonl 2> $O/r06_standalone_f64.err; cat $O/r06_standalone_f64.jsonl; tail -3 $O/r06_standalone_f64.err
rm -rf $O/prof_f64
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_f64 -o p -- python tools/bench_standalone_f64.py > /dev/null 2>&1
python tools/rocpd_summary.py $(ls $O/prof_f64/*.db | head -1) $O/r06_standalone_f64_kernel_stats.csv | head -12

#!/bin/bash
# kernel trace of one of the other BASELINE configurations:  bash tools/prof_configs.sh r02e C3
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for c in "$@"; do
  rm -rf $O/prof_${TAG}_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$c -o p -- python tools/bench_configs.py $c > $O/${TAG}_prof_$c.log 2>&1
  python tools/rocpd_summary.py $(ls $O/prof_${TAG}_$c/*.db | head -1) $O/${TAG}_kernel_stats_$c.csv | head -14
done

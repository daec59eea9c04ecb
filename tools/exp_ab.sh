#!/bin/bash
# A/B of compile-time variants of one kernel on ONE lease (the boxes of the pool differ by +-4 %):
#   gpurun -- 'SCAN_ARGS="--features ..." bash tools/exp_ab.sh "-DFLAG_A" "-DFLAG_B -DFLAG_C"'
# times tools/bench_scan.py (Mode A) with the committed build, then rebuilds libnmx.so on the box with each flag set.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
r() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms'],4), round(d['frac_of_8TBps'],4))"; }
echo "base: $(timeout 100 python tools/bench_scan.py $SCAN_ARGS 2>/dev/null | r) $(timeout 100 python tools/bench_scan.py $SCAN_ARGS 2>/dev/null | r)"
for flags in "$@"; do
  export NMX_EXTRA_CXXFLAGS="$flags"
  python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_build.log 2>&1 || tail -5 gpurun_out/exp_build.log
  echo "$flags: $(timeout 100 python tools/bench_scan.py $SCAN_ARGS 2>/dev/null | r) $(timeout 100 python tools/bench_scan.py $SCAN_ARGS 2>/dev/null | r)"
done

#!/bin/bash
# per-phase cycle counters of the time / oscillatory kernel in Mode A (C2 feature set): rebuild with -DNMX_BANK_PROFILE
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export NMX_EXTRA_CXXFLAGS="-DNMX_BANK_PROFILE"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
for spec in 1 0; do
  echo "NMX_TOW_SPEC=$spec"
  NMX_TOW_SPEC=$spec timeout 300 python tools/bench_scan.py 2>&1 | grep "timeosc profile" | sort | uniq -c | sort -rn | head -6
done
echo "one item per workgroup (NMX_TOW_PERSISTENT=0)"
NMX_TOW_PERSISTENT=0 timeout 300 python tools/bench_scan.py 2>&1 | grep "timeosc profile" | sort | uniq -c | sort -rn | head -4

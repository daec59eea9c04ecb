#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "serial schedule (NMX_OVERLAP=0):"
NMX_OVERLAP=0 timeout 300 python tools/bench_configs.py C5 2>/dev/null | grep -E "windows_per_s|\"sharp\"|\"timeosc\"|\"bank\""
echo "dense-first off:"
NMX_OVERLAP=0 NMX_SW_DENSE_FIRST=0 timeout 300 python tools/bench_configs.py C5 2>/dev/null | grep -E "windows_per_s|\"sharp\""
export NMX_EXTRA_CXXFLAGS="-DNMX_SW_PROFILE"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
NMX_OVERLAP=0 timeout 300 python tools/bench_configs.py C5 2>&1 | grep "\[sw" | sort | uniq -c | sort -rn | head -6

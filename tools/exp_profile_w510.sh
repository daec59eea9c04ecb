#!/bin/bash
# per-phase cycle counters of the 510-point time / oscillatory kernel (BASELINE config 5): rebuild with -DNMX_W510_PROFILE
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "committed build:"; timeout 300 python tools/bench_c5_split.py 2>/dev/null
export NMX_EXTRA_CXXFLAGS="-DNMX_W510_PROFILE -DNMX_SW_PROFILE"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
timeout 300 python tools/bench_c5_split.py 2>&1 | grep "\[w510" | sort | uniq -c | sort -rn | head -12
timeout 300 python tools/bench_configs.py C5 2>&1 | grep "\[sw" | sort | uniq -c | sort -rn | head -8

#!/bin/bash
# per-phase cycle counters of the sharp-wave kernel: rebuild with -DNMX_SW_PROFILE, run the FIR + sharp-wave stages alone
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export NMX_EXTRA_CXXFLAGS="-DNMX_SW_PROFILE"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
timeout 300 python tools/run_bank_only.py 2>&1 | grep "\[sw" | sort | uniq -c | sort -rn | head -12

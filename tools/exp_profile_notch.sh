#!/bin/bash
# per-phase cycle counters of the notch kernel: rebuild with -DNMX_BANK_PROFILE, run the notch alone
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/run_notch_only.py 2>&1 | tail -1
export NMX_EXTRA_CXXFLAGS="-DNMX_BANK_PROFILE"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
timeout 300 python tools/run_notch_only.py 2>&1 | grep "bank profile" | sort | uniq -c | sort -rn | head -6

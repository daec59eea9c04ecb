#!/bin/bash
# per-phase cycle counters of the FIR-bank kernel: rebuild with -DNMX_BANK_PROFILE, run the bank alone
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export NMX_EXTRA_CXXFLAGS="-DNMX_BANK_PROFILE"
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
for v in scalar rd64; do
  echo "== $v"; NMX_W64_VARIANT=$v timeout 300 python tools/run_bank_only.py 2>&1 | grep "bank" | sort | uniq -c | sort -rn | head -8
done

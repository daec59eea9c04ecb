#!/bin/bash
# profile build (per-phase counters) + normal build A/B in ONE call
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NMX_EXTRA_CXXFLAGS="-DNMX_BANK_PROFILE" python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_prof_build.log 2>&1 || tail -5 gpurun_out/exp_prof_build.log
for v in ${PROFILE_VARIANTS:-rd64}; do
  echo "== profile $v"; NMX_W64_VARIANT=$v timeout 300 python tools/run_bank_only.py 2>&1 | grep "bank" | head -4
done
python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_build.log 2>&1 || tail -5 gpurun_out/exp_build.log
bash tools/exp_variants.sh "$@"

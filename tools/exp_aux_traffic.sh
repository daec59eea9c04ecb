#!/bin/bash
# HBM fetch / write bytes of the channel-pair bank kernel for different cache-policy bits of its series stores
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in 0 2 3 17 19; do
  export NMX_EXTRA_CXXFLAGS="-DNMX_SERIES_STORE_AUX=$a"
  python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_aux_build.log 2>&1 || tail -5 gpurun_out/exp_aux_build.log
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_aux
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_aux -o p -- python tools/run_bank_only.py > /dev/null 2>&1
    echo "aux $a $c: $(python tools/rocpd_pmc.py $(ls gpurun_out/pmc_aux/*.db | head -1) | grep -A1 'bank_w64c' | tail -1)"
  done
  timeout 120 python tools/run_bank_only.py 2>/dev/null | tail -1
done

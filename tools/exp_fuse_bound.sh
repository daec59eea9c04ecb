#!/bin/bash
# What could a fused FIR-bank + Hilbert kernel for the burst bands return at most?  The bank without its band-series
# stores, the Hilbert kernel reading L2-resident series (both produce wrong results; timing only), one lease.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
h() { timeout 300 python bench.py --steps 10 --warmup 8 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; }
echo "committed build: $(h)"
echo "   serial: $(NMX_OVERLAP=0 h)"
for flags in "-DNMX_DEBUG_NO_YB" "-DNMX_DEBUG_HILBERT_SAMEROW" "-DNMX_DEBUG_NO_YB -DNMX_DEBUG_HILBERT_SAMEROW"; do
  export NMX_EXTRA_CXXFLAGS="$flags"
  python -c "import __graft_entry__ as g; g.build_lib(force=True)" > gpurun_out/exp_build.log 2>&1 || tail -5 gpurun_out/exp_build.log
  echo "$flags: $(h)"
  echo "   serial: $(NMX_OVERLAP=0 h)"
done

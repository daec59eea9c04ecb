#!/bin/bash
# waves per workgroup of the one-wave-per-item kernels (NMX_WAVES_PER_WG): headline step and the other configurations
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
h() { timeout 300 python bench.py --steps 10 --warmup 3 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; }
c() { timeout 300 python tools/bench_configs.py $1 2>/dev/null | grep -E "windows_per_s" | tr -d '\n'; echo; }
for k in 1 2 4; do
  export NMX_WAVES_PER_WG=$k
  echo "NMX_WAVES_PER_WG=$k headline: $(h)"
  echo "   serial: $(NMX_OVERLAP=0 h)"
  echo "   C3 C4 C5: $(c 'C3 C4 C5')"
done
export NMX_WAVES_PER_WG=4
timeout 900 python -m pytest tests -m gpu -q -x -k "sharp or config5 or feature_cases or alternative or pipeline or bursts" 2>&1 | tail -2

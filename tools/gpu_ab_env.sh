#!/bin/bash
# same-lease A/B of one environment knob on the headline bench: tools/gpu_ab_env.sh NMX_BANK_SPLIT_ORDER 0 1 [rounds]
K=$1; A=$2; B=$3; N=${4:-3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $K=$v timeout 300 python bench.py --steps 150 --warmup 5 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$K=$v', 'with norm', round(d['ms_per_step'], 3), 'without', round(d['ms_per_step_without_normalisation'], 3), {k: round(x, 2) for k, x in d['kernel_ms_per_step'].items()})"
  done
done

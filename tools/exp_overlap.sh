#!/bin/bash
# schedules of the step on one lease: NMX_OVERLAP = 4 (default: sharp waves on their own side stream), 2, 1, 0
cd /tmp; export TMPDIR=/tmp
for o in 4 2 4 2 1 0; do
  echo -n "NMX_OVERLAP=$o "
  NMX_OVERLAP=$o python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done

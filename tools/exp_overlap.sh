#!/bin/bash
# step time for the stream schedules (NMX_OVERLAP)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 2 4 0 2 4; do
  echo "overlap $o: $(NMX_OVERLAP=$o timeout 300 python bench.py --steps 20 --warmup 3 --cpu-windows 0 --no-cold-start 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})")"
done

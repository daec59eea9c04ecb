#!/bin/bash
# first step of a fresh plan (history fill on the workgroup threshold walk) for different workgroup sizes of that kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for nt in ${NTS:-256 128 64}; do
  for r in 1 2; do
    echo "NMX_NT_THR=$nt: $(NMX_NT_THR=$nt timeout 300 python bench.py --steps 3 --warmup 1 --cpu-windows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cold', round(d['cold_start_ms'],2), 'steady', round(d['ms_per_step'],2))")"
  done
  echo "  stream: $(NMX_NT_THR=$nt python tools/bench_stream.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print([d[k]['run_s'] for k in d])")"
done

#!/bin/bash
# default build: per-kernel times (overlapped + solo) and bit-equality of the outputs with tools/_ref_out.npy
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 6 --warmup 3 --cpu-windows 0 --no-cold-start 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'ms/step', round(d['ms_per_step'],3), round(d['value']))"
  env $cfg NMX_OVERLAP=0 timeout 300 python bench.py --steps 4 --warmup 2 --cpu-windows 0 --no-cold-start 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  solo:', {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
  env $cfg timeout 300 python tools/dump_outputs.py gpurun_out/exp/cur.npy > /dev/null 2>gpurun_out/exp/cur.err
  python -c "
import numpy as np
a=np.load('gpurun_out/exp/cur.npy'); r=np.load('tools/_ref_out.npy')
d=np.abs(a-r); print('  bit-equal to reference outputs:', np.array_equal(a,r,equal_nan=True), 'max abs diff', float(np.nanmax(d)), 'n differing', int((a!=r).sum()))"
done

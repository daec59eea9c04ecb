#!/bin/bash
# same-lease A/B of two prebuilt libraries: py_neuromodulation_amd/libnmx_prev.so (previous commit) against libnmx.so
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=py_neuromodulation_amd
h() { NMX_OVERLAP=$1 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-windows 0 --no-cold-start 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, round((d.get('roofline_modeA') or {}).get('ms_per_launch', 0), 4))"; }
cp $P/libnmx.so $P/libnmx_new.so
for rep in 1 2; do
  cp $P/libnmx_prev.so $P/libnmx.so; echo "prev: $(h 4)"; echo "   serial: $(h 0)"
  cp $P/libnmx_new.so $P/libnmx.so;  echo "new:  $(h 4)"; echo "   serial: $(h 0)"
done

#!/bin/bash
# hops per chunk of a device-resident batch (NMX_CHUNK_WINDOWS): do hand-off tensors that fit the 256 MB Infinity Cache pay
# for the smaller launches?  headline step, one lease
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
h() { timeout 300 python bench.py --steps 10 --warmup 3 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; }
for c in 1024 512 256 128 64; do echo "NMX_CHUNK_WINDOWS=$c: $(NMX_CHUNK_WINDOWS=$c h)"; done
echo "serial schedule:"
for c in 1024 256 128; do echo "NMX_CHUNK_WINDOWS=$c: $(NMX_OVERLAP=0 NMX_CHUNK_WINDOWS=$c h)"; done

#!/bin/bash
# step time for the stream schedules (NMX_OVERLAP), two rounds each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2; do for o in 2 3 1; do
  echo "overlap $o: $(NMX_OVERLAP=$o timeout 300 python bench.py --steps 20 --warmup 3 --cpu-windows 0 --no-cold-start 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done
NMX_OVERLAP=3 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "knob or default or swap" 2>&1 | tail -3
rm -rf gpurun_out/prof_ov3
NMX_OVERLAP=3 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_ov3 -o p -- python bench.py --steps 3 --warmup 2 --cpu-windows 0 --no-cold-start > /dev/null 2>&1
python tools/rocpd_timeline.py gpurun_out/prof_ov3/p_results.db 18

#!/bin/bash
# quick loop: GPU suite (-x), Mode A scan, bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; TAG=${1:-r03b}
timeout 900 python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
timeout 200 python tools/bench_scan.py > $O/${TAG}_scan.json 2>/dev/null
timeout 200 python tools/bench_scan.py --features raw_hjorth,linelength,return_raw >> $O/${TAG}_scan.json 2>/dev/null
cat $O/${TAG}_scan.json
timeout 400 python bench.py --cpu-windows 0 --no-cold-start > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"

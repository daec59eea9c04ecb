#!/bin/bash
# round 3, first GPU call: suite + bench + Mode A scan
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/r03a_pytest_gpu.log 2>&1; tail -3 $O/r03a_pytest_gpu.log
timeout 400 python bench.py > $O/r03a_bench.json 2> $O/r03a_bench.err; tail -c 1500 $O/r03a_bench.json
timeout 200 python tools/bench_scan.py > $O/r03a_scan.json 2>&1
timeout 200 python tools/bench_scan.py --features raw_hjorth,linelength,return_raw >> $O/r03a_scan.json 2>&1
cat $O/r03a_scan.json
timeout 300 python tools/bench_configs.py > $O/r03a_configs.json 2>$O/r03a_configs.err; head -c 1200 $O/r03a_configs.json

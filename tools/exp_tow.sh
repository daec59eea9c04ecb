#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 8 10 12 14 16; do echo "NMX_TOW_WAVES=$w: $(NMX_TOW_WAVES=$w timeout 100 python tools/bench_scan.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms'],3), round(d['frac_of_8TBps'],3))")"; done
echo "non-persistent: $(NMX_TOW_PERSISTENT=0 timeout 100 python tools/bench_scan.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms'],3), round(d['frac_of_8TBps'],3))")"
bash tools/pmc_scan.sh 2>&1 | tail -22

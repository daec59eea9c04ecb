#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
r() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms'],3), round(d['frac_of_8TBps'],3))"; }
echo "persistent (12 waves): $(timeout 100 python tools/bench_scan.py 2>/dev/null | r)"
echo "low1 (one item / workgroup, 5 waves per SIMD): $(NMX_TOW_LOW1=1 timeout 100 python tools/bench_scan.py 2>/dev/null | r)"
echo "non-persistent generic: $(NMX_TOW_PERSISTENT=0 timeout 100 python tools/bench_scan.py 2>/dev/null | r)"
NMX_TOW_LOW1=1 timeout 300 python -m pytest tests -m gpu -q -x -k "default or knob or modeA or c2 or random_settings" 2>&1 | tail -2

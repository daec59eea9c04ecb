#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
r() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms'],3), round(d['frac_of_8TBps'],3))"; }
echo "persistent, specialised: $(timeout 100 python tools/bench_scan.py 2>/dev/null | r)"
echo "persistent, generic (NMX_TOW_SPEC=0): $(NMX_TOW_SPEC=0 timeout 100 python tools/bench_scan.py 2>/dev/null | r)"
echo "default set (fft,welch,raw_hjorth,return_raw,linelength): $(timeout 100 python tools/bench_scan.py --features fft,welch,raw_hjorth,return_raw,linelength 2>/dev/null | r) generic: $(NMX_TOW_SPEC=0 timeout 100 python tools/bench_scan.py --features fft,welch,raw_hjorth,return_raw,linelength 2>/dev/null | r)"
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2

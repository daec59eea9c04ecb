#!/bin/bash
# sharp-wave kernel changes: parity subset, C4 / C5 shard throughput, headline step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "sharp or config5 or feature_cases or alternative or pipeline or long_windows or random_settings" 2>&1 | tail -4
timeout 300 python tools/bench_configs.py C4 C5 2>/dev/null | grep -E "windows_per_s|\"sharp\"|\"timeosc\"|^ \"C"
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"

#!/bin/bash
# FIR kernels alone and inside the step: notch alone, bank + sharp-wave filters alone, their parity tests, the step with and without overlap
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "notch alone: $(timeout 300 python tools/run_notch_only.py 2>&1 | tail -1)"
echo "bank + bank_sw + bursts + sharp alone: $(timeout 300 python tools/run_bank_only.py 2>&1 | tail -1)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "notch or bandpass or bank or preprocessing or headline or golden or pipeline or random_settings" 2>&1 | tail -2
r() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['kernel_ms_per_step'])"; }
echo "no overlap: $(NMX_OVERLAP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-mode-a 2>/dev/null | r)"
echo "overlap:    $(timeout 300 python bench.py --steps 10 --warmup 3 --no-mode-a 2>/dev/null | r)"

#!/bin/bash
# round 3, part c: register burst-statistics kernel (A/B), partitioned overlap-save FIR, long-window resampling
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "burst or headline or resampler or highrate or direct_fir or raw_resampling or preprocessing or golden or config3" 2>&1 | tail -8
r() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['kernel_ms_per_step'])"; }
for reg in 1 0; do
  echo "NMX_BURST_STAT_REG=$reg no overlap: $(NMX_BURST_STAT_REG=$reg NMX_OVERLAP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-mode-a 2>/dev/null | r)"
  echo "NMX_BURST_STAT_REG=$reg overlap:    $(NMX_BURST_STAT_REG=$reg timeout 300 python bench.py --steps 10 --warmup 3 --no-mode-a 2>/dev/null | r)"
done
rm -rf gpurun_out/prof_r3c
NMX_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3c -o p -- python bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-cold-start --no-mode-a > /dev/null 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/prof_r3c/*.db | head -1) gpurun_out/r3c_kernel_stats_nooverlap.csv | head -14
for sf in 8000 30000; do
  echo "Stream.run $sf Hz, 64 ch, 20 s: $(timeout 600 python tools/bench_stream.py --channels 64 --seconds 20 --sfreq $sf 2>&1 | tail -1)"
done | tee gpurun_out/r03_stream_high_rate.txt

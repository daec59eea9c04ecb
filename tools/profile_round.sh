#!/bin/bash
# Round profile: run on the GPU box (gpurun -- 'bash tools/profile_round.sh r01').  Writes under
# gpurun_out/; copy the summaries into profiles/ afterwards.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 400 $O/${TAG}_bench.json
NMX_OVERLAP=0 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-windows 0 > $O/${TAG}_bench_nooverlap.json 2>/dev/null
rm -rf $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o p -- python bench.py --steps 5 --warmup 2 --cpu-windows 0 > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $(ls $O/prof_$TAG/*.db | head -1) $O/${TAG}_kernel_stats.csv | head -14
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$TAG -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 > $O/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$TAG -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 > $O/${TAG}_pmc_write.log 2>&1
python tools/hbm_traffic.py $(ls $O/pmc_fetch_$TAG/*.db | head -1) $(ls $O/pmc_write_$TAG/*.db | head -1) $O/${TAG}_hbm_traffic.json
timeout 300 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu.log 2>&1; tail -2 $O/${TAG}_pytest_gpu.log
timeout 200 python tools/bench_norm.py > $O/${TAG}_norm.json 2>/dev/null; cat $O/${TAG}_norm.json

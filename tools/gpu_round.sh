#!/bin/bash
# One GPU-box call of a round: GPU parity suite (+ accepted-miss totals), smoke, bench, kernel trace, HBM-traffic PMC passes.
#   gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r03 all'
# Writes under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r03}
STEPS=${2:-all}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
if [[ $STEPS == *all* || $STEPS == *test* ]]; then
  NMX_WRITE_MISS_TOTALS=$O/${TAG}_accepted_misses.json timeout 600 python -X faulthandler -m pytest tests -m gpu -q -o faulthandler_timeout=240 > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
  cat $O/${TAG}_accepted_misses.json
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
fi
if [[ $STEPS == *all* || $STEPS == *bench* ]]; then
  timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
  tail -c 2500 $O/${TAG}_bench.json
  NMX_OVERLAP=0 timeout 300 python bench.py --steps 40 --warmup 3 --cpu-windows 0 --no-cold-start --no-mode-a > $O/${TAG}_bench_nooverlap.json 2>/dev/null
fi
if [[ $STEPS == *all* || $STEPS == *prof* ]]; then
  rm -rf $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o p -- python bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-cold-start --no-normalisation > $O/${TAG}_prof.log 2>&1
  python tools/rocpd_summary.py $(ls $O/prof_$TAG/*.db | head -1) $O/${TAG}_kernel_stats.csv | head -24
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$TAG -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-cold-start --no-normalisation > $O/${TAG}_pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$TAG -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-cold-start --no-normalisation > $O/${TAG}_pmc_write.log 2>&1
  python tools/hbm_traffic.py $(ls $O/pmc_fetch_$TAG/*.db | head -1) $(ls $O/pmc_write_$TAG/*.db | head -1) $O/${TAG}_hbm_traffic.json
fi
if [[ $STEPS == *configs* ]]; then
  timeout 600 python tools/bench_configs.py > $O/${TAG}_configs.json 2>$O/${TAG}_configs.err; cat $O/${TAG}_configs.json
fi
if [[ $STEPS == *extra* ]]; then
  bash tools/gpu_extra.sh $TAG
fi

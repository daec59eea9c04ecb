#!/bin/bash
# A/B of the feature normaliser inside the plan: scans + cells (NMX_NORM_SCAN=1, default) vs the column walk
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
for v in 1 0; do
  NMX_NORM_SCAN=$v timeout 300 python bench.py --steps 100 --warmup 5 --cpu-windows 0 --no-cold-start --no-mode-a > $O/${TAG}_norm_scan$v.json 2>$O/${TAG}_norm_scan$v.err
  python - <<PY
import json
d=json.load(open("$O/${TAG}_norm_scan$v.json"))
print("NMX_NORM_SCAN=$v", "value", round(d["value_without_normalisation"]), "ms", round(d["ms_per_step_without_normalisation"],3), "with norm", round(d["value"]), round(d["ms_per_step"],3))
PY
done
rm -rf $O/prof_norm_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_norm_$TAG -o p -- python bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-cold-start --no-mode-a > $O/${TAG}_prof_norm.log 2>&1
python tools/rocpd_summary.py $(ls $O/prof_norm_$TAG/*.db | head -1) $O/${TAG}_kernel_stats_norm.csv | grep -i "norm\|Name"; python tools/rocpd_timeline.py $(ls $O/prof_norm_$TAG/*.db | head -1) 120 2>/dev/null > $O/${TAG}_norm_timeline.txt

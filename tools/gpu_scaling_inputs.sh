#!/bin/bash
# what the strong-scaling prediction of DESIGN section 7 is built from: the headline step at 256 / N channels on one GPU
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for c in 256 128 64 32; do
  timeout 300 python bench.py --channels $c --steps 100 --warmup 5 --cpu-windows 0 --no-cold-start --no-mode-a > $O/tmp_b.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/tmp_b.json"))
print(json.dumps({"channels": $c, "ms_per_step": round(d["ms_per_step"],3), "windows_per_s": round(d["value"]), "ms_per_step_without_normalisation": round(d["ms_per_step_without_normalisation"],3), "stages": {k: round(v,3) for k,v in d["kernel_ms_per_step"].items()}}))
PY
done | tee $O/${TAG}_step_vs_channels.jsonl

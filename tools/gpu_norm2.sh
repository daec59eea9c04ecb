#!/bin/bash
# chunk length of a device-resident batch with the normaliser attached (two rounds, interleaved: box noise is +-2 %)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 150 --warmup 5 --cpu-windows 0 --no-cold-start --no-mode-a > $O/tmp_bench.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/tmp_bench.json"))
print("$label", "value", round(d["value_without_normalisation"]), "ms", round(d["ms_per_step_without_normalisation"],3), "with norm", round(d["value"]), round(d["ms_per_step"],3), "ratio", round(d["value"]/d["value_without_normalisation"],4))
PY
}
for rep in 1 2; do
for c in 128 256 384 512 1024; do
run "norm chunk $c" NMX_NORM_CHUNK_WINDOWS=$c
done
done

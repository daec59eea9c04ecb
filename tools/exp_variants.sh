#!/bin/bash
# A/B of build variants / knobs on the bench workload: per-kernel HIP-event times + output equality.
#   gpurun -- 'bash tools/exp_variants.sh "NMX_W64_VARIANT=scalar" "NMX_W64_VARIANT=rd64" ...'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp
mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== [$i] $cfg"
  env $cfg timeout 300 python bench.py --steps 6 --warmup 3 --cpu-windows 0 --no-cold-start > $O/v$i.json 2>$O/v$i.err
  python - "$O/v$i.json" << 'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, "ms/step", round(d["ms_per_step"], 3), d["kernels"]["bank"], d["kernels"]["prep"])
PY
  env $cfg NMX_OVERLAP=0 timeout 300 python bench.py --steps 4 --warmup 2 --cpu-windows 0 --no-cold-start 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  solo:', {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
  env $cfg timeout 300 python tools/dump_outputs.py $O/out$i.npy > /dev/null 2>$O/d$i.err
done
python - << 'PY'
import numpy as np, glob
fs = sorted(glob.glob("gpurun_out/exp/out*.npy"), key=lambda p: int(p.split("out")[-1].split(".")[0]))
ref = np.load(fs[0])
for f in fs[1:]:
    a = np.load(f)
    print(f, "bit-equal to out1:", np.array_equal(a, ref, equal_nan=True), "max abs diff", float(np.nanmax(np.abs(a - ref))))
PY

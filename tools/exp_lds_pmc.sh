#!/bin/bash
# LDS counters per kernel of the headline step (serial schedule): bank conflicts, LDS-array cycles, waits
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  rm -rf $O/pmc_lds
  NMX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $O/pmc_lds -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-cold-start > $O/pmc_lds.log 2>&1
  python - <<'PY'
import sqlite3, glob
from collections import defaultdict
dbs = glob.glob('gpurun_out/pmc_lds/*.db')
cur = sqlite3.connect(dbs[0]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
ix = {c: i for i, c in enumerate(cols)}
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for r in cur.execute("select * from counters_collection"):
    k = r[ix["kernel_name"]].split("(")[0]
    if "nmx_kern" not in k: continue
    acc[k][r[ix["counter_name"]]][r[ix["dispatch_id"]]] += r[ix["value"]]
for k in sorted(acc):
    print(f"{k[:48]:48s} " + "  ".join(f"{c}={sum(v.values()) / len(v):.4g}" for c, v in sorted(acc[k].items())))
PY
done

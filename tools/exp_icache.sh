#!/bin/bash
# instruction-fetch counters of the sharp-wave kernel with one and two waves per workgroup (C5 shard, serial schedule)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_CACHE|WAIT_INST|INSTS_VALU\b|INSTS_SALU\b|WAVE_CYCLES|BUSY_CYCLES|ACTIVE_INST_ANY)[A-Z_0-9]*)" | sort -u | tr '\n' ' '; echo
for k in 1 2; do
  for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH"; do
    rm -rf $O/pmc_ic
    NMX_WAVES_PER_WG=$k NMX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $O/pmc_ic -o p -- python tools/bench_configs.py C5 > $O/pmc_ic.log 2>&1
    python - "$k" <<'PY'
import sqlite3, glob, sys
from collections import defaultdict
dbs = glob.glob('gpurun_out/pmc_ic/*.db')
if not dbs:
    print(sys.argv[1], "no db"); sys.exit()
cur = sqlite3.connect(dbs[0]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
ix = {c: i for i, c in enumerate(cols)}
acc = defaultdict(lambda: defaultdict(float))
for r in cur.execute("select * from counters_collection"):
    if "sharp_dense" in r[ix["kernel_name"]]:
        acc[r[ix["counter_name"]]][r[ix["dispatch_id"]]] += r[ix["value"]]
for c, v in acc.items():
    print(f"waves/wg {sys.argv[1]}: {c} = {sum(v.values()) / len(v):.5g} per launch ({len(v)} launches)")
PY
  done
done

#!/bin/bash
# timeline (kernels + memory copies) of one page-locked host-memory batch of the headline workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof_batch
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_batch -o p -- python tools/trace_pinned_batch.py > $O/batch_trace.log 2>&1
tail -1 $O/batch_trace.log
python - <<'PY'
import sqlite3, glob
con = sqlite3.connect(glob.glob('gpurun_out/prof_batch/*.db')[0]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def t(p): return [x for x in tabs if x.startswith(p)][0]
ev = []
for r in cur.execute(f"select k.start, k.end, s.kernel_name, k.stream_id from {t('rocpd_kernel_dispatch')} k join {t('rocpd_info_kernel_symbol')} s on k.kernel_id = s.id"):
    ev.append((r[0], r[1], f"s{r[3]} " + r[2].split('(')[0][:44]))
mc = t('rocpd_memory_copy')
for r in cur.execute(f"select start, end, size from {mc}"):
    ev.append((r[0], r[1], f"copy {r[2] / 1e6:.2f} MB"))
ev.sort()
t0 = ev[0][0]
for a, b in zip(ev[:-1], ev[1:]):
    if b[0] - a[1] > 10e6:
        t0 = b[0]
for e in ev:
    if e[0] >= t0 and (e[1] - e[0] > 20e3):
        print(f"{(e[0] - t0) / 1e6:8.3f} {(e[1] - t0) / 1e6:8.3f} {(e[1] - e[0]) / 1e6:7.3f}  {e[2]}")
PY

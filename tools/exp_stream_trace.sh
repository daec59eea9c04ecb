#!/bin/bash
# timeline of one warm Stream.run (kernels + memory copies) from a rocprofv3 trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof_stream
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_stream -o p -- python tools/bench_stream.py > $O/stream_trace.log 2>&1
tail -1 $O/stream_trace.log | cut -c1-300
python - <<'PY'
import sqlite3, glob
con = sqlite3.connect(glob.glob('gpurun_out/prof_stream/*.db')[0]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def t(p): return [x for x in tabs if x.startswith(p)][0]
ev = []
for r in cur.execute(f"select k.start, k.end, s.kernel_name from {t('rocpd_kernel_dispatch')} k join {t('rocpd_info_kernel_symbol')} s on k.kernel_id = s.id"):
    ev.append((r[0], r[1], r[2].split('(')[0][:46]))
mc = t('rocpd_memory_copy')
cols = [r[1] for r in cur.execute(f"pragma table_info({mc})")]
for r in cur.execute(f"select start, end, size from {mc}"):
    ev.append((r[0], r[1], f"copy {r[2] / 1e6:.2f} MB"))
ev.sort()
# the last run: its first input copy = the first copy above 1 MB behind the last gap of more than 5 ms between events
big = [e for e in ev if e[2].startswith('copy ') and float(e[2].split()[1]) > 1.0]
t0 = big[0][0]
for a, b in zip(ev[:-1], ev[1:]):
    if b[0] - a[1] > 5e6:
        nxt = [e for e in big if e[0] >= b[0]]
        if nxt:
            t0 = nxt[0][0]
for e in ev:
    if e[0] >= t0 - 1e6 and e[0] < t0 + 40e6 and (e[1] - e[0] > 30e3):
        print(f"{(e[0] - t0) / 1e6:8.3f} {(e[1] - t0) / 1e6:8.3f} {(e[1] - e[0]) / 1e6:7.3f}  {e[2]}")
PY

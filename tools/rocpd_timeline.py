#!/usr/bin/env python
"""Kernel timeline (start offset, duration, gap to the previous kernel's end) of the LAST n dispatches in a rocprofv3
rocpd (.db) file:  python tools/rocpd_timeline.py p_results.db [n]"""
import sqlite3
import sys


def main(path, n=24):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.stream_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[-n:]
    t0 = rows[0][1]
    last_end = rows[0][1]
    for name, a, b, q in rows:
        print(f"{(a - t0) / 1e6:9.3f} ms  +{(b - a) / 1e6:7.3f} ms  gap {(a - last_end) / 1e6:7.3f}  stream {q}  {name.split('(')[0][:60]}")
        last_end = max(last_end, b)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24)

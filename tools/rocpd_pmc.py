#!/usr/bin/env python
"""Per-kernel PMC counter averages from a rocprofv3 rocpd (.db) file."""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "kernel_name" if "kernel_name" in ix else [c for c in cols if "name" in c and "kernel" in c][0]
    agg = defaultdict(lambda: defaultdict(list))
    for r in rows:
        agg[r[ix[name_c]].split("(")[0]][r[ix["counter_name"]]].append(r[ix["value"]])
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:24s} n={len(v):3d} mean={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main(sys.argv[1])

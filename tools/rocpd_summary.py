#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the per-kernel stats table that is
committed under profiles/ (same columns as `rocprofv3 --stats` kernel_stats.csv)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [])
        a.append(e - s)
    total = sum(sum(v) for v in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0]
        lines.append(f"{short},{len(v)},{sum(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)},{100 * sum(v) / total:.2f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

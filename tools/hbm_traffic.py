#!/usr/bin/env python
"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/hbm_traffic.py <fetch.db> <write.db> <out.json> [launches_to_skip]

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests as 64 B,
so bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE (both counters are in KiB).
"""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "kernel_name" if "kernel_name" in ix else [c for c in cols if "name" in c and "kernel" in c][0]
    disp_c = "dispatch_id" if "dispatch_id" in ix else None
    acc = defaultdict(lambda: defaultdict(float))
    for r in cur.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter:
            continue
        name = r[ix[name_c]].split("(")[0]
        key = r[ix[disp_c]] if disp_c else len(acc[name])
        acc[name][key] += r[ix["value"]]
    return {k: list(v.values()) for k, v in acc.items()}


def main(fetch_db, write_db, out, skip=0):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for name in sorted(set(f) | set(w)):
        if not name.startswith("nmx_") and "nmx_kern" not in name:
            continue
        fv, wv = f.get(name, [0.0])[skip:] or [0.0], w.get(name, [0.0])[skip:] or [0.0]
        fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
        res[name] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "launches": len(fv),
                     "hbm_bytes_per_launch": 2 * 1024 * fk + 1024 * wk}
    bank = next((v for k, v in res.items() if "bank_w64" in k), None)
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), averaged per launch of "
                   "bench.py's 1024-hop step. bytes = 2*1024*FETCH_SIZE + 1024*WRITE_SIZE (gfx950 FETCH "
                   "correction, MI355X_MICROARCH.md HBM section).",
           "measured_at_commit": (open(".gpurun_commit").read().strip() if __import__("os").path.exists(".gpurun_commit") else None),
           "kernels": res,
           "nmx_kern_bank_bytes_per_launch": bank["hbm_bytes_per_launch"] if bank else None}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB / launch ({v['launches']} launches)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)

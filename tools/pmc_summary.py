#!/usr/bin/env python
"""Print per-kernel averages of the PMC counters in a rocprofv3 results .db.  usage: pmc_summary.py <db> [kernel substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else [n for n in cols if "kernel" in n and "name" in n][0]
agg = {}
for r in rows:
    k = r[ix[name_col]]
    if pat not in k:
        continue
    key = (k.split("(")[0][-70:], r[ix["counter_name"]])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += r[ix["value"]]
for (k, cn), (n, tot) in sorted(agg.items()):
    print(f"{k:<72}{cn:<28}{tot / n:>18.1f}  (n={n})")

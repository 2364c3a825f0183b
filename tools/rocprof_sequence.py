#!/usr/bin/env python
"""Developer aid: the kernel sequence of ONE steady-state forward from a rocprofv3 --kernel-trace .db: the last n kernels
in start order with their durations, runs of identical consecutive (kernel, rounded duration) collapsed.
usage: rocprof_sequence.py <db> <kernels per forward>"""
import sqlite3
import sys

from rocprof_summary import short

path, n = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(path)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = sorted(c.execute(f"select {name_col}, start, end from kernels").fetchall(), key=lambda r: r[1])[-n:]
t0 = rows[0][1]
for nm, s, e in rows:
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.2f} us  {short(nm)[:110]}")
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, busy {sum(e - s for _, s, e in rows) / 1e3:.1f} us")

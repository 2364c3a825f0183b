#!/usr/bin/env python
"""Developer aid: busy time vs dispatch gaps of the LAST n kernels of a rocprofv3 --kernel-trace .db (one steady-state forward)."""
import sqlite3
import sys

from rocprof_summary import short

if __name__ == "__main__":
    path, n = sys.argv[1], int(sys.argv[2])
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = sorted(c.execute(f"select {name_col}, start, end from kernels").fetchall(), key=lambda r: r[1])[-n:]
    busy = sum(e - s for _, s, e in rows) / 1e3
    span = (rows[-1][2] - rows[0][1]) / 1e3
    gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(n - 1)]
    print(f"last {n} kernels: span {span:.1f} us, busy {busy:.1f} us, gaps {sum(gaps):.1f} us (mean {sum(gaps)/len(gaps):.2f}, max {max(gaps):.1f})")
    for nm, s, e in rows[n // 2: n // 2 + 14]:
        print(f"   {(e-s)/1e3:7.2f} us  {short(nm)[:90]}")

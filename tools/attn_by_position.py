#!/usr/bin/env python
"""Developer aid: per-position mean duration of the fused attention launches of one forward (encoder, processor layers, decoder)
from a rocprofv3 --kernel-trace .db of `ANEMOI_BENCH_SENTINEL=1 bench.py ...` (timed replays only).
usage: python tools/attn_by_position.py results.db <launches per forward>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = sorted(c.execute(f"select start, end, {name_col} from kernels").fetchall())
marks = [(s, e) for s, e, n in rows if "spin_kernel" in n and (e - s) < 200_000]
lo, hi = marks[0][1], marks[-1][0]
att = [(e - s) / 1e3 for s, e, n in rows if "gt_attn_fused_edge" in n and s >= lo and e <= hi]
steps = len(att) // per
means = [sum(att[k * per + i] for k in range(steps)) / steps for i in range(per)]
print(f"{steps} forwards; encoder {means[0]:.1f} us, processor mean {sum(means[1:-1]) / max(1, per - 2):.1f} us, decoder {means[-1]:.1f} us")

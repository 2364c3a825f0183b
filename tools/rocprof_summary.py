#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) as a per-kernel table (count, total, mean, min, max us).

    rocprof_summary.py results.db [out.txt] [--timed]

--timed keeps only the dispatches of bench.py's TIMED region: with ANEMOI_BENCH_SENTINEL=1 bench.py launches torch's spin kernel
(`torch.cuda._sleep`) right before the first and right after the last timed step; everything outside the outermost pair of spin
kernels with a short duration (warm-up, graph capture, the eager equality check, the per-kernel timing legs, which use LONG
sleeps) is dropped, so that sum(mean x calls) / steps is comparable with ms_per_step (measurement hygiene)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("anemoi::", "")
    return name[:110]


def main(path, out=None, timed=False):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    header = ""
    if timed:
        marks = sorted((s, e) for n, s, e in rows if "spin_kernel" in n and (e - s) < 200_000)  # sentinels sleep ~1 us, never 12 ms
        if len(marks) < 2:
            raise SystemExit("--timed: fewer than two sentinel (spin_kernel) dispatches; run bench.py with ANEMOI_BENCH_SENTINEL=1")
        lo, hi = marks[0][1], marks[-1][0]
        rows = [(n, s, e) for n, s, e in rows if s >= lo and e <= hi and "spin_kernel" not in n]
        header = f"# dispatches between the two sentinels only: {len(rows)} launches in {(hi - lo) / 1e3:.1f} us of wall clock\n"
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':<112}{'calls':>7}{'total_us':>12}{'mean_us':>10}{'min_us':>9}{'max_us':>9}{'pct':>7}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:<112}{a[0]:>7}{a[1]:>12.1f}{a[1]/a[0]:>10.2f}{a[2]:>9.2f}{a[3]:>9.2f}{100*a[1]/total:>7.2f}")
    lines.append(f"{'TOTAL':<112}{sum(a[0] for a in agg.values()):>7}{total:>12.1f}")
    text = header + "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if a != "--timed"]
    main(argv[0], argv[1] if len(argv) > 1 else None, timed="--timed" in sys.argv)

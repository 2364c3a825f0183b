#!/usr/bin/env python
"""Merge the per-kernel averages of the three SQ counter passes of tools/r06_pmc_sq.sh (gpurun_out/TAG/pmc_sq_<config>_pass{1,2,3}.txt) into one
table with the derived figures: cycles per launch, MFMA-busy fraction, a wave's time split, resident waves and issue occupancy per SIMD.
usage: pmc_sq_report.py DIR CONFIG [HEADER_FILE] > profiles/rNN_pmc_sq_<config>.txt"""
import collections
import re
import sys

d = collections.OrderedDict()
for i in (1, 2, 3):
    for line in open(f"{sys.argv[1]}/pmc_sq_{sys.argv[2]}_pass{i}.txt"):
        m = re.match(r"(.*?)\s+(SQ_\w+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if not m:
            continue
        k = re.sub(r"^(void )?a?nemoi::", "", m.group(1).strip()).replace("anemoi::", "")
        d.setdefault(k, collections.OrderedDict())[m.group(2)] = (float(m.group(3)), int(m.group(4)))
if len(sys.argv) > 3:
    sys.stdout.write(open(sys.argv[3]).read())
for k, v in d.items():
    print(k)
    for c, (val, n) in v.items():
        print(f"    {c:<28}{val:>16.1f}  (n={n})")
    g = lambda c: v.get(c, (0, 0))[0]  # noqa: E731
    cyc = g("SQ_BUSY_CYCLES") / 32
    if cyc:
        line = f"    -> {cyc:,.0f} cycles per launch"
        if g("SQ_VALU_MFMA_BUSY_CYCLES"):
            line += f"; MFMA pipes busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024):.3f} of the SIMD-cycles"
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            line += (f"; of a wave's time: instruction in flight {g('SQ_ACTIVE_INST_ANY') / wc:.2f}, issue stall {g('SQ_WAIT_INST_ANY') / wc:.2f}, "
                     f"parked on a wait {g('SQ_WAIT_ANY') / wc:.2f}; resident waves per SIMD {wc * 4 / (cyc * 1024):.2f}; "
                     f"SIMD issue occupancy {g('SQ_ACTIVE_INST_ANY') * 4 / (cyc * 1024):.2f} (VALU {g('SQ_ACTIVE_INST_VALU') * 4 / (cyc * 1024):.2f})")
        print(line)
    print()

#!/usr/bin/env python
"""Developer aid: time ONE kernel case of the benchmark layer with HIP events (median of `reps` single launches and the
mean of a back-to-back train).  usage: python tools/kernel_time.py <substring of the case name> [reps] [--res R]
Combine with ANEMOI_HIP_LIB=... / environment switches for same-box A/Bs."""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

res = int(sys.argv[sys.argv.index("--res") + 1]) if "--res" in sys.argv else 5
args = SimpleNamespace(data_grid="o96", hidden_res=res, layers=1, channels=512, heads=16, vars=84, kind="gt")
dev = torch.device("cuda", 0)
g, model, x = bench.build(args, dev)
model = model.to(dev).to(torch.bfloat16)
pat = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 200
with torch.inference_mode():
    cases = bench.kernel_cases(model, g, args, torch.bfloat16, dev)
    for name, (fn, bound, work) in cases.items():
        if pat not in name:
            continue
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        singles = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            singles.append(a.elapsed_time(b) * 1e3)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); b.synchronize()
        singles.sort()
        print(f"{name}: median single {singles[len(singles) // 2]:.2f} us, min {singles[0]:.2f}, train mean {a.elapsed_time(b) * 1e3 / reps:.2f} us", flush=True)

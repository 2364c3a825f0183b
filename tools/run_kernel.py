#!/usr/bin/env python
"""Developer aid: launch ONE kernel of the benchmark layer `reps` times (for rocprofv3 --pmc / --kernel-trace).
usage: python tools/run_kernel.py <substring of the case name> [reps]        (ANEMOI_RUN_KERNEL_RES=6: the res-6 hidden mesh)"""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = SimpleNamespace(data_grid="o96", hidden_res=int(os.environ.get("ANEMOI_RUN_KERNEL_RES", "5")), layers=1, channels=512, heads=16, vars=84, kind="gt")
dev = torch.device("cuda", 0)
g, model, x = bench.build(args, dev)
model = model.to(dev).to(torch.bfloat16)
pat = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
with torch.inference_mode():
    cases = bench.kernel_cases(model, g, args, torch.bfloat16, dev)
    for name, (fn, bound, work) in cases.items():
        if pat in name:
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            print("ran", name, reps, "work", work)

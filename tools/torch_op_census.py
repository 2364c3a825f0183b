#!/usr/bin/env python
"""Developer aid: which torch ops (copies, fills, casts ...) surround the HIP kernels in one eager forward, and from
which source lines.  Every one of them is a dispatch slot (~5 us when the work is tiny), which is what bounds the forward
once the mesh is sharded over several GPUs."""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    g, model, x = bench.build(args, dev)
    model = model.to(dev).to(torch.bfloat16)
    inp = {"data": x.to(dev).to(torch.bfloat16)}
    with torch.inference_mode():
        for _ in range(3):
            model(inp)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
            model(inp)
            torch.cuda.synchronize()
    cnt = Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None or (
                ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") and ev.name.startswith("aten::")):
            has_kernel = any(k for k in ev.kernels) or any(any(c.kernels) for c in ev.cpu_children)
            if not has_kernel:
                continue
            frames = [s for s in ev.stack if "anemoi_core_amd" in s or "bench.py" in s][:2]
            cnt[(ev.name, " <- ".join(f.split("/repo/")[-1] for f in frames))] += 1
    for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d}  {name:28s} {where}")
    print("total torch ops that launch kernels:", sum(cnt.values()))

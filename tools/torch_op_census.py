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
    train = os.environ.get("CENSUS_TRAIN", "0") == "1"  # census of one training step (forward + backward) instead
    model = model.to(dev).to(torch.bfloat16)
    inp = {"data": x.to(dev).to(torch.bfloat16)}
    acts = [torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]
    if train:
        model.train()

        def step():
            model.zero_grad(set_to_none=True)
            model(inp)["data"].float().square().mean().backward()
    else:
        step = lambda: model(inp)  # noqa: E731
    with torch.inference_mode(not train):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=acts, with_stack=True) as prof:
            step()
            torch.cuda.synchronize()
    cnt, dev_us = Counter(), Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None or (
                ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") and ev.name.startswith("aten::")):
            kernels, todo = [], [ev]
            while todo:  # kernels of the op and of everything it dispatches to
                e = todo.pop()
                kernels += list(e.kernels)
                todo += list(e.cpu_children)
            if not kernels:
                continue
            frames = [s for s in ev.stack if "anemoi_core_amd" in s or "bench.py" in s][:2]
            parent = ev.cpu_parent.name if ev.cpu_parent is not None else ""
            key = (ev.name, " <- ".join(f.split("/repo/")[-1] for f in frames) or parent)
            cnt[key] += 1
            dev_us[key] += sum(k.duration for k in kernels)
    for (name, where), n in sorted(cnt.items(), key=lambda kv: -dev_us[kv[0]])[:60]:
        print(f"{n:4d} {dev_us[(name, where)]:9.1f} us  {name:28s} {where}")
    print("total torch ops that launch kernels:", sum(cnt.values()), " device time", round(sum(dev_us.values())), "us")

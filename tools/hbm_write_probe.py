#!/usr/bin/env python
"""Developer aid: achievable HBM write / copy rates for buffers the size of a GEMM output (inside a hipGraph, no launch gaps)."""
import torch

for mb in (10.5, 21, 42, 84, 168):
    n = int(mb * 1e6 / 2)
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    res = []
    for name, fn in (("fill", lambda: a.fill_(1.0)), ("copy", lambda: a.copy_(b))):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10.0
        res.append(f"{name} {us:6.1f} us = {mb / us * 1e-3 * 1e3:5.2f} TB/s written")
    print(f"{mb:6.1f} MB: " + " | ".join(res))

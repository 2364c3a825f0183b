#!/usr/bin/env python
"""Developer aid: duration of ops.linear on [N, K] -> O for growing K, inside one hipGraph each (no launch gaps): the
intercept is the fixed cost of a launch (prologue, epilogue, ramp), the slope the cost per 64-wide K-step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10242
for O in (2048, 512):
    for act in (None, "gelu"):
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
            b = torch.randn(O, device="cuda").to(torch.bfloat16)
            with torch.inference_mode():
                for _ in range(3):
                    ops.linear(x, w, b, act=act)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(20):
                        y = ops.linear(x, w, b, act=act)
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) * 10.0)  # us per launch
        print(f"N={N} O={O} act={act}: " + "  ".join(f"K={k}: {t:6.1f}us" for k, t in zip((64, 128, 256, 512, 1024, 2048), row)))

#!/usr/bin/env python
"""Developer aid: device time of ops.linear over a K sweep (slope = per-K-step cost, intercept = fill + epilogue)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops  # noqa: E402


def timeit(fn, reps=20, replays=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


if __name__ == "__main__":
  N = int(sys.argv[1]) if len(sys.argv) > 1 else 10242
  for O in (2048, 512):
      for K in (64, 128, 256, 512, 1024, 2048, 4096):
          x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
          w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
          b = torch.randn(O, device="cuda").to(torch.bfloat16)
          us = timeit(lambda: ops.linear(x, w, b))
          print(f"N={N} O={O} K={K:5d}  {us:8.2f} us  {2.0*N*K*O/us/1e6:8.1f} TFLOP/s")

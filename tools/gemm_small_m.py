#!/usr/bin/env python
"""Developer aid: ops.linear on the per-rank row counts of a sharded O96 run (N = 2, 4, 8 GPUs) next to torch's GEMM."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timeit  # noqa: E402

from anemoi_core_amd import ops  # noqa: E402

for N in (1281, 2561, 5121):
    for K, O in ((512, 2048), (512, 512), (2048, 512), (512, 1024)):
        x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
        b = torch.randn(O, device="cuda").to(torch.bfloat16)
        with torch.inference_mode():
            t = timeit(lambda: ops.linear(x, w, b))
            tv = timeit(lambda: F.linear(x, w, b))
        print(f"[{N}x{K}]->{O}: ours {t:7.2f} us {2.0*N*K*O/t/1e6:7.1f} TF/s | vendor {tv:7.2f} us")

#!/usr/bin/env python
"""Developer aid: launch ops.linear on one shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py N K O [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops  # noqa: E402

N, K, O = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
b = torch.randn(O, device="cuda").to(torch.bfloat16)
for _ in range(reps):
    ops.linear(x, w, b)
torch.cuda.synchronize()

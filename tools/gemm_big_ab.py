#!/usr/bin/env python
"""Developer aid: big-tile vs 256x128 kernel on the processor GEMM shapes (set ANEMOI_GEMM_BIG=0/1 outside)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timeit  # noqa: E402

from anemoi_core_amd import ops  # noqa: E402

for N, K, O, act in ((10240, 512, 2048, "none"), (10242, 512, 2048, "none"), (10242, 512, 2048, "gelu"), (10240, 1024, 2048, "none"), (10240, 2048, 2048, "none"),
                     (10242, 2048, 512, "none"), (10242, 512, 512, "none"), (40320, 512, 2048, "none")):
    x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
    b = torch.randn(O, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.linear(x, w, b, act=(None if act == "none" else act)))
    print(f"[{N}x{K}]->{O} {act:5s} {t:7.2f} us {2.0*N*K*O/t/1e6:7.1f} TF/s")

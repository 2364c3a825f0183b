#!/usr/bin/env python
"""Developer aid: hipBLASLt / rocBLAS (torch.nn.functional.linear) next to ops.linear on the benchmark's GEMM shapes.
Run under `rocprofv3 --kernel-trace --stats` to see which vendor tile configuration was picked."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timeit  # noqa: E402

from anemoi_core_amd import ops  # noqa: E402

SHAPES = {
    "512": ((10242, 512, 2048), (10242, 512, 512), (10242, 2048, 512), (40320, 512, 2048), (40320, 2048, 512), (81840, 512, 512)),
    # the reference's default width (config/model/graphtransformer.yaml:1): q|k|v|self, MLP-1, MLP-2, projection of a processor layer and the mapper sides
    # the 40 320-row mapper sides at 512 channels: LayerNorm-fold projection (k|v, q|self), the embedding, the hidden-side projection
    "mapper": ((40320, 512, 1024), (40320, 192, 512), (40320, 128, 512), (10242, 512, 1024), (542080, 512, 1024)),
    "1024": ((10242, 1024, 4096), (10242, 4096, 1024), (10242, 1024, 1024), (40320, 1024, 2048), (40320, 1024, 4096), (40320, 4096, 1024)),
}

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "512"
    for N, K, O in SHAPES[which]:
        x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
        b = torch.randn(O, device="cuda").to(torch.bfloat16)
        with torch.inference_mode():
            t_v = timeit(lambda: F.linear(x, w, b))
            t_o = timeit(lambda: ops.linear(x, w, b))
        fl = 2.0 * N * K * O / 1e6
        print(f"[{N}x{K}]->{O}: vendor {t_v:7.2f} us {fl/t_v:7.1f} TF/s | ours {t_o:7.2f} us {fl/t_o:7.1f} TF/s")

#!/usr/bin/env python
"""Developer aid: device time of the attention forward (materialised E) and backward at the O96 processor graph size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timeit  # noqa: E402

from anemoi_core_amd import ops  # noqa: E402
from anemoi_core_amd.graphs.synthetic import build_synthetic_graph  # noqa: E402

gr = build_synthetic_graph("o8", 5)
ei = torch.from_numpy(gr.proc_edge_index).long().cuda()
n, H, C = gr.num_hidden, 16, 32
D, m = H * C, ei.shape[1]
dt = torch.bfloat16
q, k, v, g = (torch.randn(n, D, device="cuda").to(dt) for _ in range(4))
e = torch.randn(m, D, device="cuda").to(dt)
csc = ops.build_csc(ei, (n, n))
rev = ops.build_reverse_csr(csc)
out, lse = ops.gt_attention(q, k, v, e, csc, H, return_lse=True)
tf = timeit(lambda: ops.gt_attention(q, k, v, e, csc, H, return_lse=True))
tb = timeit(lambda: ops.gt_attention_backward(g, q, k, v, e, out, lse, csc, rev, H))
es = 2
fwd_bytes = es * (4 * n * D + m * D) + 4 * (m + n + 1)
bwd_bytes = es * (3 * n * D + 2 * n * D + 2 * m * D + n * D) + es * (2 * n * D + 2 * n * D) + 2 * 8 * m * H + 4 * (3 * m + 2 * n)
print(f"forward  (materialised E): {tf:7.2f} us  {fwd_bytes/tf/1e3:7.1f} GB/s of {fwd_bytes/1e6:.1f} MB compulsory")
print(f"backward (dst + src pass): {tb:7.2f} us  {bwd_bytes/tb/1e3:7.1f} GB/s of {bwd_bytes/1e6:.1f} MB compulsory")

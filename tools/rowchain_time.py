#!/usr/bin/env python
"""Developer aid: a mapper side as ONE row-chain launch (ops.gt_row_chain) against the embedding GEMM with row statistics + the LayerNorm-fold GEMM
it replaces, and the cluster chain / layer chain against each other at small row counts.  Device time of back-to-back launches in a hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timeit  # noqa: E402

from anemoi_core_amd import ops  # noqa: E402

dev, dt, D = "cuda", torch.bfloat16, 512
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
for N, K, want in ((40320, 192, False), (40320, 192, True), (10242, 64, True), (5040, 192, False), (542080, 192, True)):
    x = r(N, K).to(dt)
    we, be = (r(D, K) / K**0.5).to(dt), (0.1 * r(D)).to(dt)
    gm, bt = (1 + 0.2 * r(D)).to(dt), (0.1 * r(D)).to(dt)
    wq, bq = (r(1024, D) / 22).to(dt), (0.1 * r(1024)).to(dt)
    wqg, dq = ops.fold_layer_norm(wq, bq, gm, bt)
    vec = torch.cat([be.float(), dq]).to(dt).contiguous()
    wef, wqf = ops.pack_embedding_frag(we), ops.pack_weight_frag(wqg)
    ws = (wq.float() * gm.float()).to(dt).contiguous()
    c, d = ws.float().sum(1).contiguous(), (wq.float() @ bt.float() + bq.float()).contiguous()

    def two():
        y, st = ops.linear_with_row_stats(x, we, be)
        return ops.linear_ln_folded(y, ws, c, d, st, 1e-5)

    with torch.inference_mode():
        t1 = timeit(lambda: ops.gt_row_chain(x, wef, wqf, vec, 1024, 1e-5, want_x_out=want))
        t2 = timeit(two) if K % 64 == 0 else float("nan")
    print(f"rows {N:6d} K {K:3d} x_out {int(want)}: row chain {t1:7.2f} us | embedding GEMM + LayerNorm-fold GEMM {t2:7.2f} us")

#!/usr/bin/env python
"""Developer aid: device time of decoder-type gt_chain2 launches (no trailing projection, or a narrow one; several panel rounds) - back-to-back
launches in a hipGraph.  Combine with ANEMOI_HIP_LIB=... for same-box A/Bs.  usage: python tools/chain2_tail_time.py [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timeit  # noqa: E402

from anemoi_core_amd import ops  # noqa: E402

dev, dt, D, HD = "cuda", torch.bfloat16, 512, 2048
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
rows = [int(a) for a in sys.argv[1:]] or [10242, 40320, 542080]
for N in rows:
    for q_out, want_x in ((0, True), (128, False), (2048, True)):
        attn, x = r(N, D).to(dt), r(N, D).to(dt)
        wp, bp = (r(D, D) / 22).to(dt), (0.1 * r(D)).to(dt)
        g1, be1 = (1 + 0.2 * r(D)).to(dt), (0.1 * r(D)).to(dt)
        w1, b1 = (r(HD, D) / 22).to(dt), (0.1 * r(HD)).to(dt)
        w2, b2 = (r(D, HD) / 45).to(dt), (0.1 * r(D)).to(dt)
        w1g, d1 = ops.fold_layer_norm(w1, b1, g1, be1)
        parts = [bp.float(), d1, b2.float()]
        wqg = None
        if q_out:
            wq, bq = (r(q_out, D) / 22).to(dt), (0.1 * r(q_out)).to(dt)
            wq_, dq = ops.fold_layer_norm(wq, bq, g1, be1)
            wqg = ops.pack_weight_frag(wq_)
            parts.append(dq)
        vec = torch.cat(parts).to(dt).contiguous()
        wpf, w1f, w2f = ops.pack_weight_frag(wp), ops.pack_weight_frag(w1g), ops.pack_weight_frag(w2)
        with torch.inference_mode():
            t = timeit(lambda: ops.gt_layer_chain2(attn, x, wpf, w1f, w2f, vec, HD, 1e-5, wqg=wqg, q_out_features=q_out, lnq_eps=1e-5, want_x_out=want_x))
        panels = (N + 47) // 48
        rounds = (panels + 255) // 256
        print(f"rows {N:6d} q_out {q_out:4d} x_out {int(want_x)}: {t:8.2f} us  ({panels} panels, {rounds} rounds: {t / rounds:6.2f} us per round)")

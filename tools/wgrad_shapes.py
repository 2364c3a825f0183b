#!/usr/bin/env python
"""Developer aid: census of the weight-gradient GEMM shapes of one O96 training step and their isolated device time
(transposes + split-K GEMM), to see where the backward spends its GEMM time."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from anemoi_core_amd import autograd as ag  # noqa: E402

if __name__ == "__main__":
    args = bench.parse()
    dev = torch.device("cuda", 0)
    g, model, x = bench.build(args, dev)
    model = model.to(dev).to(torch.bfloat16).train()
    inp = {"data": x.to(dev).to(torch.bfloat16)}
    shapes = collections.Counter()
    orig = ag._weight_grad

    def spy(dz, xx):
        if not ag._tn_ok(dz, xx):
            shapes[(dz.shape[0], dz.shape[1], xx.shape[1])] += 1
        return orig(dz, xx)

    from anemoi_core_amd import ops

    orig_tn = ops.linear_wgrad

    def spy_tn(dz, xx, with_bias_grad=False):
        shapes[(dz.shape[0], dz.shape[1], xx.shape[1])] += 1
        return orig_tn(dz, xx, with_bias_grad)

    ag._weight_grad = spy
    ops.linear_wgrad = spy_tn
    model(inp)["data"].float().square().mean().backward()
    ag._weight_grad = orig
    ops.linear_wgrad = orig_tn
    torch.cuda.synchronize()
    total = 0.0
    print(f"{'rows':>8} {'out':>6} {'in':>6} {'calls':>5} {'us/call':>9} {'TFLOP/s':>8} {'ms/step':>8}")
    for (n, o, k), c in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
        dz = torch.randn(n, o, device=dev, dtype=torch.bfloat16)
        xx = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            orig(dz, xx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            orig(dz, xx)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        total += us * c / 1e3
        print(f"{n:8d} {o:6d} {k:6d} {c:5d} {us:9.1f} {2.0 * n * o * k / us / 1e6:8.1f} {us * c / 1e3:8.2f}")
    print(f"total {total:.2f} ms per step")

#!/usr/bin/env python
"""Developer aid: host cost of an EAGER launch through the binding in use (ANEMOI_TORCH_EXT=1 TORCH_LIBRARY layer, =0 ctypes), per
grad context: torch.inference_mode, torch.no_grad with parameters that require grad (what a module's weights are), torch.no_grad
with plain tensors.  Small operands, so the device is never the bound: the figure is microseconds of host time per call.  Also
times the O96 model's eager forward in the first two contexts."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from anemoi_core_amd import ops  # noqa: E402


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    x = torch.randn(256, 512, device=dev).to(torch.bfloat16)
    w = torch.randn(512, 512, device=dev).to(torch.bfloat16)
    b = torch.randn(512, device=dev).to(torch.bfloat16)
    g = torch.ones(512, device=dev).to(torch.bfloat16)
    wp, bp, gp = (torch.nn.Parameter(t.clone()) for t in (w, b, g))
    print(f"ANEMOI_TORCH_EXT={os.environ.get('ANEMOI_TORCH_EXT', '1')}")
    for name, ctx, (W, B, G) in (("inference_mode", torch.inference_mode, (w, b, g)), ("no_grad, parameters", torch.no_grad, (wp, bp, gp)),
                                 ("no_grad, plain tensors", torch.no_grad, (w, b, g))):
        with ctx():
            lin = per_call(lambda: ops.linear(x, W, B))
            ln = per_call(lambda: ops.layer_norm(x, G, B))
        print(f"  {name:24s} linear {lin:6.1f} us   layer_norm {ln:6.1f} us per call (host)")
    args = bench.parse()
    _, model, xin = bench.build(args, dev)
    model = model.to(dev).to(torch.bfloat16).eval()
    inp = {"data": xin.to(dev).to(torch.bfloat16)}
    for name, ctx, train in (("inference_mode", torch.inference_mode, False), ("no_grad", torch.no_grad, False),
                             ("no_grad, model.train()", torch.no_grad, True)):
        model.train(train)
        with ctx():
            for _ in range(3):
                model(inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model(inp)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"  O96 model, eager forward under {name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms (host issue {(t1 - t0) / 10 * 1e3:.2f} ms)")

#!/usr/bin/env python
"""Probe: do two HALF-SIZE chains of the processor's GEMMs on two streams (one hipGraph with two branches) overlap each
other's fixed costs (launch, cold first tile, output drain)?  Compares with ONE chain on all rows.
usage: python tools/two_stream_probe.py [rows] [layers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
torch.manual_seed(0)
bf = torch.bfloat16
W = {k: (torch.randn(o, i, device=dev) * i ** -0.5).to(bf) for k, (o, i) in
     dict(qkvs=(2048, 512), proj=(512, 512), m1=(2048, 512), m2=(512, 2048)).items()}
B = {k: torch.zeros(w.shape[0], device=dev, dtype=bf) for k, w in W.items()}


def chain(x, layers):
    for _ in range(layers):
        qkvs = ops.linear(x, W["qkvs"], B["qkvs"])
        o = ops.linear(qkvs[:, :512], W["proj"], B["proj"], residual=x)
        h = ops.linear(o, W["m1"], B["m1"], act="gelu")
        x = ops.linear(h, W["m2"], B["m2"], residual=o)
    return x


def timed(fn, n=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


with torch.inference_mode():
    x = torch.randn(N, 512, device=dev).to(bf)
    xa, xb = x[: N // 2].contiguous(), x[N // 2:].contiguous()
    t_full = timed(lambda: chain(x, L))
    t_half = timed(lambda: chain(xa, L))

    s2 = torch.cuda.Stream()

    def two():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            chain(xb, L)
        chain(xa, L)
        cur.wait_stream(s2)

    t_two = timed(two)
    print(f"rows {N}, {L} layers x 4 GEMMs: one chain on all rows {t_full:.1f} us; one chain on half the rows {t_half:.1f} us; "
          f"two half chains on two streams {t_two:.1f} us")

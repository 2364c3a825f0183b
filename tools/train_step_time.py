#!/usr/bin/env python
"""Developer aid: forward+backward of the O96 GraphTransformer model (bf16, one GPU) through the training path; also checks
that every parameter receives a finite gradient at full size."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    args = bench.parse()
    dev = torch.device("cuda", 0)
    g, model, x = bench.build(args, dev)
    model = model.to(dev).to(torch.bfloat16).train()
    inp = {"data": x.to(dev).to(torch.bfloat16)}

    def step():
        model.zero_grad(set_to_none=True)
        out = model(inp)["data"]
        out.float().square().mean().backward()
        return out

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    bad = [k for k, p in model.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    with torch.no_grad():
        for _ in range(3):  # the first forward after an optimiser-free training loop re-derives the packed / folded weights
            model(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(inp)
        torch.cuda.synchronize()
        ms_f = (time.perf_counter() - t0) / n * 1e3
    # whole step (forward + backward) as one hipGraph: static input / gradient buffers
    ms_g = float("nan")
    try:
        if os.environ.get("TRAIN_GRAPH", "1") != "1":
            raise RuntimeError("graph leg disabled")
        def gstep():  # gradients are (re)allocated from the graph's pool at fixed addresses: no zero fills
            model.zero_grad(set_to_none=True)
            out = model(inp)["data"]
            out.float().square().mean().backward()

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                gstep()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gstep()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        ms_g = (time.perf_counter() - t0) / n * 1e3
    except Exception as e:  # noqa: BLE001
        import traceback

        traceback.print_exc()
    print(f"train step as one hipGraph: {ms_g:.2f} ms")
    print(f"train step (fwd+bwd, eager, bf16): {ms:.2f} ms; inference forward (eager): {ms_f:.2f} ms; parameters without a finite gradient: {len(bad)} {bad[:5]}")
    print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")

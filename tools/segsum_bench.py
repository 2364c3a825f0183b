#!/usr/bin/env python
"""Developer aid: `anemoi_edge_ln_residual_segment_sum_fwd` alone at the GNN shapes (bf16, 512 channels): the processor mesh
(81 840 edges -> 10 242 destinations), an encoder-like graph (many edges per destination) and a decoder-like one (3 per destination);
microseconds per launch as a hipGraph of 20 launches, GB/s on the algorithmic bytes (3 x M x 1 KiB + N x 1 KiB) and the largest
deviation from a torch fp32 evaluation.  Eight operand sets in turn, so that every launch reads rows that are not in the Infinity Cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops  # noqa: E402
from anemoi_core_amd.layers.graphcache import get_csc  # noqa: E402


def reference(z, e, g, b, eps, dst, n):
    y = torch.nn.functional.layer_norm(z.float(), (z.shape[1],), g.float(), b.float(), eps).to(z.dtype).float() + e.float()
    y = y.to(z.dtype)
    return y, torch.zeros(n, z.shape[1], device=z.device).index_add_(0, dst.long(), y.float())


def run(name, ei, n_dst, dev):
    D = 512
    M = ei.shape[1]
    csc = get_csc(ei, (int(ei[0].max()) + 1, n_dst), True)
    sets = [(torch.randn(M, D, device=dev).to(torch.bfloat16), torch.randn(M, D, device=dev).to(torch.bfloat16)) for _ in range(8)]
    z, e = sets[0]
    g = (1 + 0.1 * torch.randn(D, device=dev)).to(torch.bfloat16)
    b = (0.1 * torch.randn(D, device=dev)).to(torch.bfloat16)
    en, agg = ops.edge_ln_residual_segment_sum(z, e, g, b, 1e-5, csc)
    rn, ragg = reference(z, e, g, b, 1e-5, ei[1], n_dst)
    d_e = (en.float() - rn.float()).abs().max().item()
    d_a = (agg.float() - ragg).abs().max().item()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.edge_ln_residual_segment_sum(z, e, g, b, 1e-5, csc)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):  # 8 operand sets (8 x 3 x M KiB > the 256 MB Infinity Cache) in turn: every launch reads cold rows
            for i in range(24):
                ops.edge_ln_residual_segment_sum(sets[i % 8][0], sets[i % 8][1], g, b, 1e-5, csc)
    for _ in range(3):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 240
    mb = (3 * M + n_dst) * D * 2 / 1e6
    print(f"  {name:10s} M={M:7d} N={n_dst:6d}: {us:7.2f} us  {mb / us:5.2f} TB/s   max|d e_new| {d_e:.3g}  max|d agg| {d_a:.3g} (bf16 rounding of sums up to {ragg.abs().max().item():.1f})")


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    n = 10242
    # processor-like: ~8 in-edges per destination, ragged (6..36), dst-sorted
    deg = torch.randint(5, 12, (n,))
    deg[:42] = 36
    dst = torch.repeat_interleave(torch.arange(n), deg)
    ei = torch.stack([torch.randint(0, n, (dst.numel(),)), dst]).to(dev)
    run("processor", ei, n, dev)
    deg = torch.randint(10, 20, (n,))
    dst = torch.repeat_interleave(torch.arange(n), deg)
    ei = torch.stack([torch.randint(0, 40320, (dst.numel(),)), dst]).to(dev)
    run("encoder", ei, n, dev)
    dst = torch.repeat_interleave(torch.arange(40320), 3)
    ei = torch.stack([torch.randint(0, n, (dst.numel(),)), dst]).to(dev)
    run("decoder", ei, 40320, dev)

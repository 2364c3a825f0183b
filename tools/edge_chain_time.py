#!/usr/bin/env python
"""Device time of the GNN edge chain (csrc/gnn_chain.hip) at the O96 processor's size, back to back, warm clocks.
ANEMOI_EDGE_CHAIN_DBG (bit 0: no GELU) / ANEMOI_CHAIN_ROWS vary the kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops
dev, dt, D = "cuda", torch.bfloat16, 512
M, N = int(os.environ.get("M", 81840)), 10242
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
e, p = r(M, D).to(dt), r(N, 2 * D).to(dt)
dst = torch.sort(torch.randint(0, N, (M,), device=dev, generator=g)).values.to(torch.int32)
src = torch.randint(0, N, (M,), device=dev, generator=g).to(torch.int32)
P = ops.pack_weight_frag
w = [P((r(D, D) / 22).to(dt)) for _ in range(3)]
b = [(0.1 * r(D)).to(dt) for _ in range(3)]
gam, bet = torch.ones(D, device=dev, dtype=dt), torch.zeros(D, device=dev, dtype=dt)
f = lambda: ops.gnn_edge_chain(e, p[:, :D], dst, p[:, D:], src, w[0], b[0], w[1], b[1], w[2], b[2], gam, bet, 1e-5)
for _ in range(300):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    f()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"edge chain M={M}: {us:.1f} us per launch = {2.0 * M * 3 * D * D / us / 1e6:.0f} TFLOP/s (dbg={os.environ.get('ANEMOI_EDGE_CHAIN_DBG', '0')}, rows={os.environ.get('ANEMOI_CHAIN_ROWS', 'auto')})")

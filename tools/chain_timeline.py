#!/usr/bin/env python
"""EXPERIMENTS BUILD ONLY (python -m anemoi_core_amd.build --experiments; ANEMOI_HIP_LIB=anemoi_core_amd/lib/libanemoi_hip_exp.so).
In-kernel timeline of the round-4 row-resident layer chain (csrc/experiments/gt_chain.hip, instrumented instantiation): shader-clock stamps of
wave 0 at every phase boundary of each workgroup's first panel, as medians over the workgroups, in microseconds.

    python tools/chain_timeline.py [--rows 10242] [--no-q]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from experiments_ops import gt_layer_chain  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10242)
ap.add_argument("--no-q", action="store_true")
ap.add_argument("--rows-per-tile", type=int, default=0)
args = ap.parse_args()
dev, dt, D, HD, N = "cuda", torch.bfloat16, 512, 2048, args.rows
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
attn, x = r(N, D).to(dt), r(N, D).to(dt)
wp, w1, w2, wq = (r(D, D) / 22).to(dt), (r(HD, D) / 22).to(dt), (r(D, HD) / 45).to(dt), (r(4 * D, D) / 22).to(dt)
bp, b1, b2, bq = r(D).to(dt) * 0.1, r(HD).to(dt) * 0.1, r(D).to(dt) * 0.1, r(4 * D).to(dt) * 0.1
g1, be1 = torch.ones(D, device=dev, dtype=dt), torch.zeros(D, device=dev, dtype=dt)
P = ops.pack_weight_frag
kw = {} if args.no_q else dict(lnq_w=g1, lnq_b=be1, wq=P(wq), bq=bq)
call = lambda tl=None: gt_layer_chain(attn, x, P(wp), bp, g1, be1, 1e-5, P(w1), b1, P(w2), b2, rows_per_tile=args.rows_per_tile, timeline=tl, **kw)  # noqa: E731
for _ in range(400):  # ~50 ms of work: the clocks of an idle GPU take milliseconds to ramp (a cold launch runs at about half speed)
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
print(f"plain kernel: {e0.elapsed_time(e1) * 1e3 / 20:.2f} us per launch (back to back, {N} rows)")
tl = torch.zeros(256, 8, 48, dtype=torch.int64, device=dev)
for _ in range(100):
    call(tl)
torch.cuda.synchronize()
e0.record()
for _ in range(20):  # back to back: one bracketed launch would include ~50 us of launch latency
    call(tl)
e1.record()
torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) * 1e3 / 20
tw = tl.cpu().double()  # [wg, wave, slot]
n = int((tw[0, 0] > 0).sum())
nwg = int((tw[:, 0, 0] > 0).sum())
tw = tw[:nwg, :, :n]
t = tw[:, 0]
span = (tw[:, :, n - 1].max(1).values - tw[:, :, 0].min(1).values).median().item()
mhz = span / wall_us  # shader clock ticks per us, calibrated on the launch's wall time (upper bound of the true rate)
names = ["entry", "loads issued", "panel rows arrived", "panel in LDS", "proj GEMM", "x1 + LN"]
for c in range(4):
    names += [f"MLP-1[{c}] GEMM", f"GELU[{c}] in regs", f"barrier[{c}]"] + ([f"MLP-2[{c - 1}] GEMM"] if c else [])
names += ["barrier[end]", "MLP-2[3] GEMM", "x2 epilogue"]
if not args.no_q:
    names += ["LN'"] + [f"qkvs[{c}] GEMM+store" for c in range(4)]
print(f"instrumented launch {wall_us:.1f} us, {n} stamps, median workgroup span {span:.0f} ticks -> {mhz:.0f} ticks/us")
prev = t[:, 0]
for i in range(1, n):
    d = ((t[:, i] - prev) / mhz)
    print(f"  {names[i] if i < len(names) else i:24s} +{d.median().item():6.2f} us   (min {d.min().item():5.2f}, max {d.max().item():5.2f})   at {((t[:, i] - t[:, 0]) / mhz).median().item():7.2f}")
    prev = t[:, i]

print("per wave: median over workgroups of (stamp - the workgroup's first stamp) in us; waves 0-7")
base = tw[:, :, 0].min(1, keepdim=True).values
for i in range(n):
    row = ((tw[:, :, i] - base) / mhz).median(0).values
    print(f"  {names[i] if i < len(names) else i:24s} " + " ".join(f"{v:7.2f}" for v in row.tolist()))

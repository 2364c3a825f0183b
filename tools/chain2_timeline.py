#!/usr/bin/env python
"""The layer chain (csrc/gt_chain2.hip): time per launch back to back (with the EXPERIMENTS build of the library - python -m
anemoi_core_amd.build --experiments, ANEMOI_HIP_LIB=anemoi_core_amd/lib/libanemoi_hip_exp.so - also next to the round-4 chain,
csrc/experiments/gt_chain.hip, on the same inputs), and - experiments build only - the in-kernel timeline of the instrumented instantiation - shader-clock stamps of every wave at every step
boundary of each workgroup's first panel, medians over the workgroups, in microseconds (group A = waves 0-3, group B = waves 4-7).

    python tools/chain2_timeline.py [--rows 10242] [--no-q] [--no-timeline]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import _lib, ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from experiments_ops import gt_layer_chain  # noqa: E402

HAVE_EXPERIMENTS = hasattr(_lib.load(), "anemoi_gt_chain_fwd")

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10242)
ap.add_argument("--no-q", action="store_true")
ap.add_argument("--no-timeline", action="store_true")
ap.add_argument("--rows-per-tile", type=int, default=0)
ap.add_argument("--interleave", default="none", choices=["none", "add", "attn"], help="another kernel between the launches, as in the model: a torch elementwise add over 84 MB / a [10242 x 2048] GEMM of this library")
ap.add_argument("--layers", type=int, default=16, help="distinct weight sets cycled through, as the layers of a model do (1: the same weights every launch)")
args = ap.parse_args()
dev, dt, D, HD, N = "cuda", torch.bfloat16, 512, 2048, args.rows
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
attn, x = r(N, D).to(dt), r(N, D).to(dt)
wp, w1, w2, wq = (r(D, D) / 22).to(dt), (r(HD, D) / 22).to(dt), (r(D, HD) / 45).to(dt), (r(4 * D, D) / 22).to(dt)
bp, b1, b2, bq = r(D).to(dt) * 0.1, r(HD).to(dt) * 0.1, r(D).to(dt) * 0.1, r(4 * D).to(dt) * 0.1
g1, be1 = (1 + 0.2 * r(D)).to(dt), (0.1 * r(D)).to(dt)
P = ops.pack_weight_frag
kw1 = {} if args.no_q else dict(lnq_w=g1, lnq_b=be1, wq=P(wq), bq=bq)
w1g, d1 = ops.fold_layer_norm(w1, b1, g1, be1)
wqg, dq = ops.fold_layer_norm(wq, bq, g1, be1)
qf = 0 if args.no_q else 4 * D
vec = torch.cat([bp.float(), d1, b2.float()] + ([] if args.no_q else [dq])).to(dt).contiguous()
L = max(1, args.layers)
sets = []
for i in range(L):  # the same values in distinct memory: what differs between the layers of a model is where the weights live, not what they hold
    c = lambda t: None if t is None else t.clone()  # noqa: E731
    sets.append(dict(wp=c(P(wp)), w1=c(P(w1)), w2=c(P(w2)), wq=None if args.no_q else c(P(wq)), w1g=c(P(w1g)), wqg=None if args.no_q else c(P(wqg)), vec=c(vec)))
it1, it2 = [0], [0]


def v1():
    w = sets[it1[0] % L]
    it1[0] += 1
    kw1 = {} if args.no_q else dict(lnq_w=g1, lnq_b=be1, wq=w["wq"], bq=bq)
    return gt_layer_chain(attn, x, w["wp"], bp, g1, be1, 1e-5, w["w1"], b1, w["w2"], b2, rows_per_tile=args.rows_per_tile, **kw1)


def v2(tl=None):
    w = sets[it2[0] % L]
    it2[0] += 1
    return ops.gt_layer_chain2(attn, x, w["wp"], w["w1g"], w["w2"], w["vec"], HD, 1e-5, wqg=w["wqg"], q_out_features=qf, rows_per_tile=args.rows_per_tile, timeline=tl)


big = torch.zeros(N, 4096, device=dev, dtype=dt)
wl = (r(2048, D) / 22).to(dt)


def between():
    if args.interleave == "add":
        big.add_(1.0)
    elif args.interleave == "attn":
        ops.linear(x, wl, None)


def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
        between()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for _ in range(300):  # the clocks of an idle GPU take milliseconds to ramp
    v2()
torch.cuda.synchronize()
if HAVE_EXPERIMENTS:
    a, b = v1(), v2()
    a, b = (a, b) if args.no_q else (a[0], b[0])
    print(f"v2 against v1: max |dx2| {float((a.float() - b.float()).abs().max()):.4f} at scale {float(a.float().abs().max()):.2f}")
for rep in range(3):
    t1 = timed(v1) if HAVE_EXPERIMENTS else float("nan")
    t2 = timed(v2)
    print(f"{N} rows, {L} weight sets, back to back: round-4 chain {t1:7.2f} us   role-split chain {t2:7.2f} us per launch")
if args.no_timeline or not HAVE_EXPERIMENTS:
    sys.exit(0)
tl = torch.zeros(256, 8, 48, dtype=torch.int64, device=dev)
for _ in range(50):
    v2(tl)
torch.cuda.synchronize()
wall_us = timed(lambda: v2(tl))
tw = tl.cpu().double()  # [wg, wave, slot]
nwg = int((tw[:, 0, 0] > 0).sum())
na, nb = int((tw[0, 0] > 0).sum()), int((tw[0, 4] > 0).sum())
span = (tw[:nwg].max(2).values.max(1).values - tw[:nwg, :, 0].min(1).values).median().item()
mhz = span / wall_us
print(f"instrumented launch {wall_us:.1f} us, {na} / {nb} stamps (group A / B), median workgroup span {span:.0f} ticks -> {mhz:.0f} ticks/us")
hc, qc = HD // D, qf // D
names_a = ["entry", "S0 rows requested", "S0 all requested", "S0 rows stored", "S1 start (panel in LDS)", "S1 P GEMM (64 columns)", "S1 x1 -> bufC + stats", "S2 LN -> bufB"]
for t in range(hc):
    names_a += [f"M{t} start", f"M{t} M1 GEMM", f"M{t} GELU -> h"]
names_a += [f"M{hc} (idle) start", f"M{hc} passed", "S8 x2 -> global"]
for k in range(0, qc, 2):
    names_a += [f"Q{k} start", f"Q{k} GEMM", f"Q{k} stores issued"]
names_b = ["entry", "S0 skip rows, vectors, ring", "S1 start (panel in LDS)", "S1 P GEMM (64 columns)", "S1 x1 -> bufC + stats", "S2 LN -> bufB", "S2 acc2 = b2 + x1", "M0 (idle) start"]
for t in range(1, hc + 1):
    names_b += [f"M{t} start", f"M{t} M2 GEMM"] + (["x2 -> buf + stats"] if t == hc else [])
names_b += ["S8 LN' -> bufB"]
for k in range(1, qc, 2):
    names_b += [f"Q{k} start", f"Q{k} GEMM", f"Q{k} stores issued"]
base = tw[:nwg, :, 0].min(1, keepdim=True).values
for grp, names, n, w0 in (("A", names_a, na, 0), ("B", names_b, nb, 4)):
    print(f"group {grp}: median over workgroups of (stamp - the workgroup's first stamp) in us, waves {w0}-{w0 + 3}; delta of the group's first wave")
    prev = None
    for i in range(n):
        row = ((tw[:nwg, w0:w0 + 4, i] - base) / mhz).median(0).values
        d = "" if prev is None else f"  (+{row[0].item() - prev:5.2f})"
        prev = row[0].item()
        print(f"  {names[i] if i < len(names) else i:24s} " + " ".join(f"{v:7.2f}" for v in row.tolist()) + d)

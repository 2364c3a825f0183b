#!/usr/bin/env python
"""In-kernel timeline of the GraphConv edge chain (csrc/gnn_chain.hip, instrumented instantiation): shader-clock stamps of all 8 waves
at the phase boundaries of a workgroup's first five 64-row panels, as medians over the workgroups, in microseconds.

    python tools/edge_chain_timeline.py [--rows 81840]

(The instrumented build spills 14 registers - the residual's row addresses across the third GEMM, reloaded in the last epilogue: that
epilogue reads ~1 us long here; the GEMM segments carry no spill code.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=81840)
args = ap.parse_args()
dev, dt, D, M, N = "cuda", torch.bfloat16, 512, args.rows, 10242
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
e, p = r(M, D).to(dt), r(N, 2 * D).to(dt)
dst = torch.sort(torch.randint(0, N, (M,), device=dev, generator=g)).values.to(torch.int32)
src = torch.randint(0, N, (M,), device=dev, generator=g).to(torch.int32)
P = ops.pack_weight_frag
w = [P((r(D, D) / 22).to(dt)) for _ in range(3)]
b = [(0.1 * r(D)).to(dt) for _ in range(3)]
gam, bet = torch.ones(D, device=dev, dtype=dt), torch.zeros(D, device=dev, dtype=dt)
plain = lambda: ops.gnn_edge_chain(e, p[:, :D], dst, p[:, D:], src, w[0], b[0], w[1], b[1], w[2], b[2], gam, bet, 1e-5)  # noqa: E731
out = torch.empty_like(e)
n_wg = min(256, (M + 63) // 64)
tl = torch.zeros(n_wg, 8, 48, dtype=torch.int64, device=dev)
g1, g2 = p[:, :D], p[:, D:]
lib = _lib.load()


def timed():
    _lib.check(lib.anemoi_gnn_edge_chain_timeline(e.data_ptr(), D, g1.data_ptr(), 2 * D, dst.data_ptr(), g2.data_ptr(), 2 * D, src.data_ptr(),
                                                  w[0].data_ptr(), b[0].data_ptr(), w[1].data_ptr(), b[1].data_ptr(), w[2].data_ptr(), b[2].data_ptr(),
                                                  gam.data_ptr(), bet.data_ptr(), 1e-5, out.data_ptr(), D, M, tl.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "edge_chain_timeline")


def per_launch(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for _ in range(300):  # warm clocks
    plain()
torch.cuda.synchronize()
print(f"plain kernel: {per_launch(plain):.1f} us per launch (back to back, {M} rows)")
for _ in range(50):
    timed()
wall_us = per_launch(timed)
if os.environ.get("ANEMOI_EDGE_CHAIN_DBG", "0") == "0":
    assert torch.equal(out, plain()), "the instrumented instantiation must compute what the production one does"
t = tl.cpu().double()  # [wg, wave, slot]
n = int((t[0, 0] > 0).sum())
span = (t[:, :, :n].amax((1, 2)) - t[:, :, 0].amin(1)).median().item()
mhz = span / wall_us  # ticks per us, calibrated on the launch's wall time (an upper bound of the true rate)
print(f"instrumented launch {wall_us:.1f} us, {n} stamps per wave, median workgroup span {span:.0f} ticks -> {mhz:.0f} ticks/us")
names = ["e panel ready", "GEMM 1", "gather-add + GELU -> h1 (barrier)", "GEMM 2 (+ next panel)", "GELU -> h2 in place (2 barriers)", "GEMM 3",
         "bias + row statistics (barrier)", "LayerNorm + residual + stores"]
rel = (t[:, :, :n] - t[:, :1, :1]) / mhz  # us since the workgroup's wave-0 entry
npan = (n - 1) // 8
print("phase durations, median over workgroups of (latest wave's stamp - latest wave's previous stamp), us; columns = panels of a workgroup")
last = rel.amax(1)  # [wg, slot]: the moment the LAST wave passed the stamp
print(f"  {'(first panel in LDS)':38s}" + f"{(last[:, 1] - last[:, 0]).median().item():8.2f}")
for ph in range(1, 8):
    row = [(last[:, 1 + 8 * pn + ph] - last[:, 1 + 8 * pn + ph - 1]).median().item() for pn in range(npan)]
    print(f"  {names[ph]:38s}" + "".join(f"{v:8.2f}" for v in row))
row = [(last[:, 1 + 8 * (pn + 1)] - last[:, 8 * (pn + 1)]).median().item() for pn in range(npan - 1)]
print(f"  {'end-of-panel barrier':38s}" + "".join(f"{v:8.2f}" for v in row))
tot = [(last[:, 8 * (pn + 1)] - last[:, 1 + 8 * pn]).median().item() for pn in range(npan)]
print(f"  {'panel total':38s}" + "".join(f"{v:8.2f}" for v in tot))
print("per wave (panel 2 of each workgroup): median of (stamp - the workgroup's panel-2 start), us; waves 0-7 (0-3 = first wave of SIMD 0-3)")
if npan >= 2:
    base = rel[:, :, 1 + 8].amin(1, keepdim=True)
    for ph in range(8):
        vals = (rel[:, :, 1 + 8 + ph] - base).median(0).values
        print(f"  {names[ph]:38s}" + "".join(f"{v.item():8.2f}" for v in vals))

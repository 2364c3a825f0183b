#!/usr/bin/env python
"""Developer aid: where does the time of one big-tile GEMM launch go?  Needs the instrumented build
(`bash tools/build_alt.sh timing -DANEMOI_DBG_TIMING` on a copy of csrc/linear.hip with the DBG_T marks; see git history) loaded
through ANEMOI_HIP_LIB.  Prints, averaged over the 256 workgroups, the wall-clock offsets (100 MHz constant clock) of: kernel
entry, first DMAs issued, first K-tile landed, second K-step, end of the K-loop, epilogue start, last store issued, stores done."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import _lib, ops  # noqa: E402

N, K, O = 10242, 512, 2048
act = sys.argv[1] if len(sys.argv) > 1 else None
x = torch.randn(N, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(O, K, device="cuda") / K**0.5).to(torch.bfloat16)
b = torch.randn(O, device="cuda").to(torch.bfloat16)
lib = _lib.load()
fn = lib.anemoi_debug_read_timing
fn.argtypes = [C.c_void_p]
fn.restype = C.c_int
rows = []
for rep in range(8):
    ops.linear(x, w, b, act=None if act in (None, "none") else act)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(256, 8).astype(np.float64)
    t0 = t[:, 0].min()
    rows.append(((t - t0) / 100.0))  # us
t = np.stack(rows[2:]).mean(0)
names = ["entry", "DMA issued", "K-tile 0 landed", "K-step 1", "K-loop done", "epilogue start", "last store issued", "stores done"]
for i, n in enumerate(names):
    print(f"{n:20s} mean {t[:, i].mean():7.2f} us   min {t[:, i].min():7.2f}   max {t[:, i].max():7.2f}")

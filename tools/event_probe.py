"""How do HIP event pairs behave around single kernels when the queue is idle vs backed up?  (bench.py profile_forward)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anemoi_core_amd import ops

dev = "cuda"
x = torch.randn(10242, 1024, device=dev, dtype=torch.bfloat16)
w = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16) * 0.03
b = torch.zeros(1024, device=dev, dtype=torch.bfloat16)
for _ in range(3000):
    ops.linear(x, w, b)
torch.cuda.synchronize()
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record()
for _ in range(500):
    ops.linear(x, w, b)
a1.record(); torch.cuda.synchronize()
print("back-to-back us/launch:", a0.elapsed_time(a1) * 1e3 / 500)


def run(preroll, n=12):
    ev = []
    if preroll:
        torch.cuda._sleep(int(preroll))
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear(x, w, b); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return [round(a.elapsed_time(c) * 1e3, 1) for a, c in ev], [round(ev[i][1].elapsed_time(ev[i + 1][0]) * 1e3, 1) for i in range(n - 1)]


for pre in (0, 2_000_000, 0, 8_000_000, 0):
    d, gaps = run(pre)
    print("preroll" if pre else "idle   ", "pairs us:", d, "\n         gaps us:", gaps)

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_core_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep import timeit
N = 10242
for D in (512, 2048):
    x = torch.randn(N, D, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    g = torch.ones(D, device="cuda", dtype=torch.bfloat16)
    mb = N * D * 2 / 1e6
    t = timeit(lambda: ops.layer_norm(x, g, g)); print(f"D={D} layernorm  {t:7.2f} us  {2*mb/t*1e-3*1e3:7.0f} GB/s (r+w)")
    t = timeit(lambda: y.zero_()); print(f"D={D} zero_      {t:7.2f} us  {mb/t*1e3:7.0f} GB/s (w)")
    t = timeit(lambda: y.copy_(x)); print(f"D={D} copy_      {t:7.2f} us  {2*mb/t*1e3:7.0f} GB/s (r+w)")
    t = timeit(lambda: x.sum()); print(f"D={D} sum        {t:7.2f} us  {mb/t*1e3:7.0f} GB/s (r)")

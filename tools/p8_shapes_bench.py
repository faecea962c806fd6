"""Isolated timing of the persistent GEMM on the step's NT shapes with their epilogues (for A/B of launch-time knobs
-- BIFOLD_P8_GROUP_M, BIFOLD_P8_NT, BIFOLD_P8_REVK ... -- the library reads them once per process: run once per setting)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("vis qkv", 50432, 2304, 768, 0), ("vis fc1", 50432, 3072, 768, L.ACT_QUICK_GELU), ("vis out", 50432, 768, 768, 0),
          ("vis fc2", 50432, 768, 3072, 0), ("fus qkv", 70400, 1536, 512, 0), ("fus fc1", 70400, 2048, 512, L.ACT_ERF_GELU),
          ("fus out", 70400, 512, 512, 0), ("fus fc2", 70400, 512, 2048, 0)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = 0.0
for name, m, n, k, act in shapes:
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev, generator=g).float()
    t = timeit(lambda: ops.gemm(a, b, bias=bias, act=act))
    tot += t
    print(f"{name:8s} {m}x{n}x{k}: {t*1e3:7.1f} us {2*m*n*k/t/1e9:5.0f} TF/s", flush=True)
print(f"sum {tot*1e3:.1f} us  ({' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('BIFOLD_'))})")

"""Isolated timing of the persistent GEMM's WRITE-HEAVY epilogues on the step's shapes (fp32 result + residual, the
fc1 + GELU' pair, QuickGELU fc1): the launches BIFOLD_P8_SKEW acts on.  One setting per process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
hdt = torch.float16
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("fus out+res", 70400, 512, 512, dict(res=1)), ("fus fc2+res", 70400, 512, 2048, dict(res=1)),
          ("fus fc1+g'", 70400, 2048, 512, dict(act=L.ACT_ERF_GELU_SAVEGRAD, want_pre=True)),
          ("vis fc1", 50432, 3072, 768, dict(act=L.ACT_QUICK_GELU)), ("vis qkv", 50432, 2304, 768, dict())]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = 0.0
for name, m, n, k, o in shapes:
    a = torch.randn(m, k, device=dev, generator=g).to(hdt)
    b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(hdt)
    kw = dict(bias=torch.randn(n, device=dev, generator=g))
    if o.get("act"): kw["act"] = o["act"]
    if o.get("want_pre"): kw["want_pre"] = True
    if o.get("res"):
        kw["residual"] = torch.randn(m, n, device=dev, generator=g); kw["out_dtype"] = torch.float32
    t = min(timeit(lambda: ops.gemm(a, b, **kw)) for _ in range(3))
    tot += t
    print(f"{name:12s} {t*1e3:7.1f} us {2*m*n*k/t/1e9:5.0f} TF/s", end=" | ")
print(f"sum {tot*1e3:.1f} us  ({os.environ.get('BIFOLD_P8_SKEW', 'no skew')})")

"""What the vendor library (hipBLASLt through torch.matmul / F.linear) reaches on the step's GEMM shapes -- a ceiling
estimate for the hand-written kernels, NOT part of the product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("vis qkv", 50432, 2304, 768), ("vis fc1", 50432, 3072, 768), ("vis out", 50432, 768, 768),
          ("vis fc2", 50432, 768, 3072), ("fus qkv", 70400, 1536, 512), ("fus fc1", 70400, 2048, 512),
          ("fus out", 70400, 512, 512), ("fus fc2", 70400, 512, 2048), ("txt qkv", 19712, 1536, 512),
          ("txt fc1", 19712, 2048, 512), ("big", 8192, 8192, 8192)]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev, generator=g).bfloat16()
    t_lt = timeit(lambda: torch.nn.functional.linear(a, b))
    t_ltb = timeit(lambda: torch.nn.functional.linear(a, b, bias))
    t_own = timeit(lambda: ops.gemm(a, b))
    fl = 2 * m * n * k
    print(f"{name:8s} {m}x{n}x{k}: hipBLASLt {t_lt*1e3:7.1f} us {fl/t_lt/1e9:5.0f} TF/s | +bias {t_ltb*1e3:7.1f} us {fl/t_ltb/1e9:5.0f} | own {t_own*1e3:7.1f} us {fl/t_own/1e9:5.0f} TF/s", flush=True)

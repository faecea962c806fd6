"""e4m3 tower GEMMs (LayerNorm output x weights) isolated: persistent kernel vs the 128 x 128 one vs the bf16 persistent GEMM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, m, n, k in [("vis qkv", 50432, 2304, 768), ("vis fc1", 50432, 3072, 768), ("txt qkv", 19712, 1536, 512),
                      ("txt fc1", 19712, 2048, 512), ("vis qkv B512", 100864, 2304, 768), ("vis fc1 B512", 100864, 3072, 768)]:
    x = torch.randn(m, k, device=dev, generator=g)
    w = torch.randn(n, k, device=dev, generator=g) * k ** -0.5
    bias = torch.randn(n, device=dev, generator=g)
    a8, sa = ops.quantize_rows_fp8(x)
    b8, sb = ops.quantize_rows_fp8(w)
    xb, wb = x.bfloat16(), w.bfloat16()
    t8 = timeit(lambda: ops.gemm_fp8(a8, sa, b8, sb, bias=bias))
    t16 = timeit(lambda: ops.gemm(xb, wb, bias=bias))
    fl = 2 * m * n * k
    print(f"{name:13s} {m}x{n}x{k}: e4m3 {t8*1e3:7.1f} us {fl/t8/1e9:5.0f} TF/s | bf16 {t16*1e3:7.1f} us {fl/t16/1e9:5.0f} TF/s", flush=True)

"""In-kernel cycle probe of the persistent e4m3 GEMM (-DP8_PROBE build)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fn = L.lib.bf_debug_p8_probe
fn.restype = ctypes.c_int
for name, m, n, k in [("vis qkv", 50432, 2304, 768), ("vis fc1", 50432, 3072, 768), ("deep", 50432, 768, 3072)]:
    x = torch.randn(m, k, device=dev, generator=g)
    w = torch.randn(n, k, device=dev, generator=g) * k ** -0.5
    a8, sa = ops.quantize_rows_fp8(x)
    b8, sb = ops.quantize_rows_fp8(w)
    f = lambda: ops.gemm_fp8(a8, sa, b8, sb)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf)
    print(f"{name} {m}x{n}x{k}: {ms*1e3:.1f} us {2*m*n*k/ms/1e9:.0f} TF/s | main {buf[0]} cyc / {buf[2]} pairs = {buf[0]/max(buf[2],1):.0f} per K=128 pair (MFMA issue 1024), clock {buf[0]/max(buf[1],1)*100:.0f} MHz | epilogue+prologue {buf[3]} | next-item wait {buf[6]}", flush=True)

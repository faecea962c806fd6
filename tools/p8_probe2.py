import ctypes, sys, os, torch
sys.path.insert(0, "/root/repo")
from bifold_amd.hip import ops, lib as L
L.lib.bf_gemm_set_variant(8)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fn = L.lib.bf_debug_p8_probe
fn.restype = ctypes.c_int
m, n, k = 70400, 2048, 512
a = torch.randn(m, k, device=dev, generator=g).bfloat16()
b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
bkn = (torch.randn(k, n, device=dev, generator=g) * k ** -0.5).bfloat16()
bias = torch.randn(n, device=dev, generator=g)
aux = torch.randn(m, n, device=dev, generator=g).bfloat16()
cases = {
    "fc1 bias+erf+pre": lambda: ops.gemm(a, b, bias=bias, act=L.ACT_ERF_GELU, want_pre=True),
    "fc1 bias+erf": lambda: ops.gemm(a, b, bias=bias, act=L.ACT_ERF_GELU),
    "plain": lambda: ops.gemm(a, b),
    "nn act3 aux": lambda: ops.gemm(a, bkn, b_kn=True, act=L.ACT_ERF_GELU_GRAD, aux=aux),
    "nn plain": lambda: ops.gemm(a, bkn, b_kn=True),
}
pad = 64
outp = torch.empty(m, n + pad, device=dev, dtype=torch.bfloat16)[:, :n]
bknp = torch.empty(k, n + pad, device=dev, dtype=torch.bfloat16)[:, :n]
bknp.copy_(bkn)
auxp = torch.empty(m, n + pad, device=dev, dtype=torch.bfloat16)[:, :n]
auxp.copy_(aux)
cases["plain, ldc padded"] = lambda: ops.gemm(a, b, out=outp)
cases["nn plain, ldb+ldc padded"] = lambda: ops.gemm(a, bknp, b_kn=True, out=outp)
cases["nn act3, all padded"] = lambda: ops.gemm(a, bknp, b_kn=True, act=L.ACT_ERF_GELU_GRAD, aux=auxp, out=outp)
cases["nn plain, only ldb padded"] = lambda: ops.gemm(a, bknp, b_kn=True)
for name, f in cases.items():
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf)
    print(f"{name:18s}: {ms*1e3:.1f} us {2*m*n*k/ms/1e9:.0f} TF/s | main {buf[0]} cyc ({buf[0]/max(buf[2],1):.0f}/stage) clock {buf[0]/max(buf[1],1)*100:.0f} MHz; prologue issue {buf[5]}, epilogue {buf[3]-buf[5]}, next wait {buf[6]}")

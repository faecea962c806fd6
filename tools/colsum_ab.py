import sys, torch
sys.path.insert(0, "/root/repo")
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
m, n, k = 70400, 2048, 512
a = torch.randn(m, k, device=dev, generator=g).bfloat16()
bkn = (torch.randn(k, n, device=dev, generator=g) * k ** -0.5).bfloat16()
aux = torch.randn(m, n, device=dev, generator=g).bfloat16()
cs = torch.zeros(n, device=dev)
def t(f, n_it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_it * 1e3
for rep in range(2):
    print("fused   :", t(lambda: ops.gemm(a, bkn, b_kn=True, act=L.ACT_ERF_GELU_GRAD, aux=aux, colsum_out=cs)), "us")
    def sep():
        o = ops.gemm(a, bkn, b_kn=True, act=L.ACT_ERF_GELU_GRAD, aux=aux)
        ops.colsum(o, out=cs)
    print("separate:", t(sep), "us")
    print("gemm only:", t(lambda: ops.gemm(a, bkn, b_kn=True, act=L.ACT_ERF_GELU_GRAD, aux=aux)), "us")

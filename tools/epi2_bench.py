"""The dX GEMM of the fusion MLP with the saved-GELU' multiplier (NN, 70 400 x 2 048 x 512, EPI 2) against its plain form."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
m, n, k = 70400, 2048, 512
for hdt in (torch.float16, torch.bfloat16):
    dy = torch.randn(m, k, device=dev, generator=g).to(hdt)
    w2 = (torch.randn(k, n, device=dev, generator=g) * 0.05).to(hdt)
    aux = torch.rand(m, n, device=dev, generator=g).to(hdt)
    cs = torch.empty(n, device=dev)
    def timed(name, fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{str(hdt):16s} {name:34s} {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us", flush=True)
    timed("plain", lambda: ops.gemm(dy, w2, b_kn=True))
    timed("x saved gelu' (MUL_AUX)", lambda: ops.gemm(dy, w2, b_kn=True, act=L.ACT_MUL_AUX, aux=aux))
    timed("x saved gelu' + bias sums", lambda: ops.gemm(dy, w2, b_kn=True, act=L.ACT_MUL_AUX, aux=aux, colsum_out=cs))

import sys, torch
sys.path.insert(0, "/root/repo")
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
for (m, n, k, res, odt) in [(50432, 768, 768, True, torch.float32), (50432, 768, 3072, True, torch.float32),
                            (70400, 512, 2048, True, torch.float32), (50432, 768, 768, False, torch.bfloat16),
                            (50000, 776, 256, True, torch.float32)]:
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev, generator=g)
    r = torch.randn(m, n, device=dev, generator=g) if res else None
    ref = a.float() @ b.float().t() + bias
    if res:
        ref = ref + r
    for env in ("tail", "notail"):
        out = ops.gemm(a, b, bias=bias, residual=r, out_dtype=odt)
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(4):
            e0.record()
            for _ in range(3):
                ops.gemm(a, b, bias=bias, residual=r, out_dtype=odt)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        print(f"{m}x{n}x{k} res={res} {odt}: rel err {err:.2e}  {min(ts)*1e3:.1f} us  {2*m*n*k/min(ts)/1e9:.0f} TF/s", flush=True)
        break

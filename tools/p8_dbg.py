import sys, torch
sys.path.insert(0, "/root/repo")
from bifold_amd.hip import ops, lib as L
L.lib.bf_gemm_set_variant(8)
dev = torch.device("cuda:0")
for (m, n, k, odt) in [(512, 768, 768, torch.float32), (512, 768, 768, torch.bfloat16), (394, 768, 768, torch.float32), (256, 256, 64, torch.float32), (256, 256, 32, torch.float32), (8192, 768, 768, torch.float32)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(m, k, generator=g).bfloat16().to(dev)
    b = torch.randn(n, k, generator=g).bfloat16().to(dev)
    ref = a.float() @ b.float().t()
    out = ops.gemm(a, b, out_dtype=odt, split_k=1).float()
    err = (out - ref).abs()
    print(m, n, k, odt, "max err", err.max().item(), "ref max", ref.abs().max().item())
    bad = (err > 0.05 * ref.abs().max()).nonzero()
    if len(bad):
        print("  bad count", len(bad), "first", bad[:5].tolist(), "rows range", bad[:, 0].min().item(), bad[:, 0].max().item(), "cols", bad[:, 1].min().item(), bad[:, 1].max().item())
        r, c = bad[0].tolist()
        print("  out", out[r, c].item(), "ref", ref[r, c].item())

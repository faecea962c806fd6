import sys, os, torch
sys.path.insert(0, os.getcwd())
from bifold_amd.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
L, H, D = 275, 16, 32
for B in (16, 32, 64, 128, 256, 512):
    qkv = torch.randn(B * L, 3 * H * D, device=dev, generator=g).half()
    dout = torch.randn(B * L, H * D, device=dev, generator=g).half()
    out, lse = ops.attention_fwd(qkv, B, L, H, D, D ** -0.5, 0, None, need_lse=True)
    for tag, fn in (("fwd", lambda: ops.attention_fwd(qkv, B, L, H, D, D ** -0.5, 0, None, need_lse=True)),
                    ("bwd", lambda: ops.attention_bwd(qkv, out, dout, lse, B, L, H, D, D ** -0.5, 0, None))):
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1) / 5)
        ts.sort(); t = ts[len(ts)//2] * 1e3
        print(f"B={B:4d} {tag}: {t:8.1f} us   {t / (B * H) * 1e3:7.2f} ns per (batch, head)   x256/B = {t * 256 / B:7.1f} us")

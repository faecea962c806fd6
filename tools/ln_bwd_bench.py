"""LayerNorm backward at the fusion's shape (70 400 x 512), fp32 gradient stream and the fp16 one (R16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rows, cols = 70400, 512
x = torch.randn(rows, cols, device=dev, generator=g)
gamma, beta = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
h, mean, rstd = ops.layernorm_fwd(x, gamma, beta, torch.float16, save_stats=True)
dy = torch.randn(rows, cols, device=dev, generator=g).half()
dres32 = torch.randn(rows, cols, device=dev, generator=g)
dres16 = dres32.half()
dg, db, s1, s2 = (torch.empty(cols, device=dev) for _ in range(4))
def timed(name, fn, mb):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:44s} {us:6.1f} us  {mb / us:5.2f} TB/s of {mb:.0f} MB (both launches: ln_bwd + colsum_multi)", flush=True)
timed("ln_bwd fp32 stream, with bias sums", lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres32, torch.float16, dg, db, s1, s2), 576)
timed("ln_bwd fp32 stream", lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres32, torch.float16, dg, db), 576)
timed("ln_bwd fp16 stream, with bias sums", lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres16, torch.float16, dg, db, s1, s2, stream16=True), 360)
timed("ln_bwd fp16 stream", lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres16, torch.float16, dg, db, stream16=True), 360)

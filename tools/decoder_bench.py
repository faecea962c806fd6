"""Isolated timings of the decoder / loss kernels at the bench shape (B = 256, 196 patches, width 512, two heads):
the stretch between the last fusion layer's forward and the first backward GEMM, where nothing else runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops
from bifold.models.decoder_math import upsample_operator
dev = torch.device("cuda:0")
B, grid, dim, nh = 256, 14, 512, 2
L = grid * grid + 1
g = torch.Generator(device=dev).manual_seed(0)
tok = torch.randn(B, L, dim, device=dev, generator=g)
w_eff = torch.randn(nh, dim, device=dev, generator=g) * 0.05
b_eff = torch.zeros(nh, device=dev)
U = upsample_operator(grid).to(dev)
S = grid * 16
target = torch.rand(nh, B, S, S, device=dev, generator=g)
d_tok = torch.empty_like(tok)
g_w, g_b = torch.empty(nh, dim, device=dev), torch.empty(nh, device=dev)
one = torch.ones(1, device=dev)
def timed(name, fn, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:34s} {us:7.1f} us   {bytes_ / us / 1e6:5.2f} TB/s of {bytes_ / 1e6:.0f} MB", flush=True)
low = ops.decoder_project(tok, L * dim, w_eff, b_eff, B, nh, grid * grid, dim)
heat = ops.decoder_upsample_sigmoid(low, U, nh * B, grid).view(nh, B, S, S)
d_heat = [ops.bce_mean_bwd(heat[h], target[h], one) for h in range(nh)]
g_low = torch.empty(nh, B, grid * grid, device=dev)
map_b = B * S * S * 4
timed("decoder_project", lambda: ops.decoder_project(tok, L * dim, w_eff, b_eff, B, nh, grid * grid, dim), tok.numel() * 4)
timed("upsample_sigmoid (both heads)", lambda: ops.decoder_upsample_sigmoid(low, U, nh * B, grid), nh * map_b)
timed("bce_mean (one head)", lambda: ops.bce_mean(heat[0], target[0]), 2 * map_b)
timed("bce_mean_bwd (one head)", lambda: ops.bce_mean_bwd(heat[0], target[0], one), 3 * map_b)
timed("upsample_sigmoid_bwd (one head)", lambda: ops.decoder_upsample_sigmoid_bwd(heat[0], d_heat[0], U, B, grid, out=g_low[0]), 2 * map_b)
timed("decoder_project_bwd (2 launches)", lambda: ops.decoder_project_bwd(tok, L * dim, w_eff, g_low, d_tok, L * dim, g_w, g_b, B, nh, grid * grid, dim), 2 * tok.numel() * 4)

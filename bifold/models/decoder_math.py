"""Host-side constants of the collapsed heatmap decoder.

The reference's ConvDecoder (bifold/models/modules.py:58-71) interleaves its 1x1 convs
with four `nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)`.  Each
upsample is a fixed linear map that acts separably on rows and columns, so four of them
compose into one matrix U [16 g, g]; the HIP kernels apply U on both sides.
"""
from __future__ import annotations

import torch


def bilinear_x2_matrix(n: int) -> torch.Tensor:
    """[2n, n] float64 matrix of one 1-D bilinear x2 upsample, align_corners=False
    (source coordinate 0.5*(i+0.5)-0.5, clamped below at 0, right neighbour clamped to n-1)."""
    m = torch.zeros(2 * n, n, dtype=torch.float64)
    for i in range(2 * n):
        src = max(0.5 * (i + 0.5) - 0.5, 0.0)
        i0 = min(int(src), n - 1)
        i1 = min(i0 + 1, n - 1)
        lam = src - i0
        m[i, i0] += 1.0 - lam
        m[i, i1] += lam
    return m


def upsample_operator(grid: int, levels: int = 4) -> torch.Tensor:
    """U [grid * 2**levels, grid] as float32 (composed in float64); rows sum to 1."""
    u = torch.eye(grid, dtype=torch.float64)
    n = grid
    for _ in range(levels):
        u = bilinear_x2_matrix(n) @ u
        n *= 2
    # the device kernels sum over the band |j - x // 2**levels| <= 1 only (include/bifold_hip.h)
    x = torch.arange(grid * 2 ** levels).unsqueeze(1) // 2 ** levels
    j = torch.arange(grid).unsqueeze(0)
    if bool((u[(x - j).abs() > 1] != 0).any()):
        raise RuntimeError("upsample operator is not banded: the decoder kernels would drop terms")
    return u.to(torch.float32).contiguous()

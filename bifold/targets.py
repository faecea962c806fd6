"""Training-target synthesis on the device (reference bifold/data/processor.py:289-297, 405-439).

The reference builds every target heatmap in numpy on a DataLoader worker (one fp64
meshgrid + exp per head per sample) and ships 4 x 200 KB per bimanual sample over PCIe.
Here only the annotated pixels travel ([B, heads, P, 2] floats) and the maps are written
straight into HBM by bf_gaussian_targets, bit-for-bit the reference's first / gmm maps up to
the last ulp of fp64 exp.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Sequence

import torch

from bifold_amd.hip import ops

HEADS = {False: ("pick", "place"), True: ("left_pick", "left_place", "right_pick", "right_place")}


def make_gaussmaps(points: torch.Tensor, counts: Optional[torch.Tensor], image_size: int, sigma: float = 5.0,
                   strategy: str = "gmm") -> torch.Tensor:
    """Batched Processor._make_gaussmap: points fp32 [n, P, 2] (x, y) pixels on the GPU, counts
    int32 [n] valid points per map (0 -> the all-zero map the reference uses for an absent arm,
    processor.py:296) -> fp32 [n, S, S]."""
    return ops.gaussian_targets(points.float(), counts, int(image_size), float(sigma), strategy)


def heatmap_targets(pixels: Mapping[str, torch.Tensor], image_size: int, sigma: float = 5.0,
                    strategy: str = "gmm", counts: Optional[Mapping[str, torch.Tensor]] = None,
                    heads: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
    """{head: [B, P, 2] or [B, 2] pixels} -> {f"{head}_heatmap": [B, S, S]} in ONE launch."""
    heads = list(heads or pixels)
    pts = [pixels[h].float().reshape(pixels[h].shape[0], -1, 2) for h in heads]
    p = max(t.shape[1] for t in pts)
    if any(t.shape[1] != p for t in pts):
        raise ValueError("every head must carry the same number of point slots (pad and use counts)")
    b = pts[0].shape[0]
    stacked = torch.stack(pts, 0).reshape(len(heads) * b, p, 2)
    cnt = None
    if counts is not None:
        cnt = torch.stack([counts[h].to(torch.int32) for h in heads], 0).reshape(-1).contiguous()
    maps = make_gaussmaps(stacked, cnt, image_size, sigma, strategy).view(len(heads), b, image_size, image_size)
    return {f"{h}_heatmap": maps[i] for i, h in enumerate(heads)}

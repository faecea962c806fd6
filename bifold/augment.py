"""Spatial augmentation of a training batch on the device (reference bifold/data/processor.py:302-354,
`Processor._spatial_augmentation` / `_aug_pixel`), batched: the reference runs it per sample on a DataLoader
worker -- a random (angle, dx, dy), the annotated pixels rotated about the centre of the (S - 1)-sized frame and
shifted, up to `max_augmentation_trials` redraws until every pixel stays inside, then the same warp applied to
every rgb / depth image of the sample (`v2.functional.affine`, nearest neighbour).

Here the few hundred bytes of pixels are handled on the host (numpy RNG and the in-frame test, as in the
reference) and the images -- 150 KB to 600 KB each -- are warped where they live, by bf_affine_nearest.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from bifold_amd.hip import ops


def aug_pixel(pixel: torch.Tensor, angle: float, dx: float, dy: float, size: float) -> Tuple[torch.Tensor, bool]:
    """processor.py:344-354 without the assert: pixels [k, 2] (x, y) -> (rotated by `angle` degrees about
    size / 2 and shifted by (dx, dy), whether all of them are inside [0, size))."""
    rad = np.deg2rad(angle)
    rot = torch.from_numpy(np.array([[np.cos(rad), -np.sin(rad)], [np.sin(rad), np.cos(rad)]])).float()
    p = pixel.float() - size / 2
    p = p @ rot.T
    p = p + size / 2
    p[:, 0] += dx
    p[:, 1] += dy
    return p, bool(torch.all(p >= 0) and torch.all(p < size))


def draw_augmentation(pixels: Mapping[str, Optional[torch.Tensor]], image_size: int, rotate=(-5, 6), translate=(-5, 6),
                      max_trials: int = 5, rng=np.random):
    """One sample: ({head: augmented [k, 2] pixels}, (angle, dx, dy)) of the first trial that keeps every
    annotated pixel in frame, or (None, None) when none does (the reference then leaves the sample untouched).
    The pixels rotate by -angle while the images are warped with +angle (processor.py:313-316, 333-341), exactly
    as the reference writes it; the translation is common to both."""
    for _ in range(max_trials):
        angle = rng.uniform(*rotate)
        dx = rng.uniform(*translate)
        dy = rng.uniform(*translate)
        out, ok = {}, True
        for k, val in pixels.items():
            if val is None:
                continue
            out[k], inside = aug_pixel(val.clone().reshape(-1, 2), -angle, dx, dy, size=image_size - 1)
            if not inside:
                ok = False
                break
        if ok:
            return out, (angle, dx, dy)
    return None, None


def augment_batch(sample: Dict[str, torch.Tensor], image_size: int, cfg=None, rng=np.random) -> Dict[str, torch.Tensor]:
    """Batched Processor._spatial_augmentation: every `*pick*` / `*place*` pixel tensor ([B, 2] or [B, k, 2], on any
    device) and every `rgb*` / `depth*` image ([B, C, S, S] or [B, T, C, S, S] on the GPU) of `sample`, one draw
    per sample.  Samples whose trials all leave the frame keep the identity warp (angle = dx = dy = 0)."""
    rotate = tuple(cfg["rotate_augmentation"]) if cfg else (-5, 6)
    translate = tuple(cfg["translate_augmentation"]) if cfg else (-5, 6)
    trials = int(cfg["max_augmentation_trials"]) if cfg else 5
    heads = [k for k in sample if ("pick" in k or "place" in k) and "heatmap" not in k and torch.is_tensor(sample[k])]
    images = [k for k in sample if ("rgb" in k or "depth" in k) and torch.is_tensor(sample[k]) and sample[k].dim() >= 4
              and sample[k].is_floating_point()]
    if not images:
        raise ValueError("augment_batch: the sample holds no rgb / depth image")
    batch = sample[images[0]].shape[0]
    host = {k: sample[k].detach().float().cpu().clone() for k in heads}     # never write into the caller's tensors
    params = np.zeros((batch, 3), dtype=np.float32)
    for b in range(batch):
        moved, p = draw_augmentation({k: host[k][b] for k in heads}, image_size, rotate, translate, trials, rng)
        if moved is None:
            continue
        params[b] = p
        for k, v in moved.items():
            host[k][b] = v.reshape(host[k][b].shape)
    out = dict(sample)
    for k in heads:
        out[k] = host[k].to(sample[k].device)
    dev = sample[images[0]].device
    par = torch.from_numpy(params).to(dev)
    for k in images:
        x = sample[k]
        if x.dim() == 5:                            # context frames [B, T, C, S, S]: one warp per sample, all frames
            b, t = x.shape[:2]
            out[k] = ops.affine_nearest(x.reshape(b * t, *x.shape[2:]).float(),
                                        par.repeat_interleave(t, dim=0)).view_as(x)
        else:
            out[k] = ops.affine_nearest(x.float(), par)
    out["augmentation"] = par
    return out


def process_depth(depth: torch.Tensor, mask: Optional[torch.Tensor], cfg=None, train: bool = True,
                  rng=np.random) -> torch.Tensor:
    """Processor._process_depth (reference bifold/data/processor.py:59-78, 122-124) for a batch of depth frames that
    are already at the model's resolution: DepthScale in training when `depth_augmentations.random_depth_shift`
    (one uniform draw per frame from numpy's generator, like the reference's per-sample call), MaskDepth when
    `mask_depth`, float32, TruncatedDepthStandardization when `standardize_depth` -- two launches on the device
    (bf_depth_prepare, bf_depth_standardize).  depth / mask: fp32 [B, 1, S, S] or [B, S, S] in HBM.
    `add_depth_noise` is open3d's Redwood noise simulator (data/depth_augmentations.py:20-32), host-side third-party
    code outside this build: asked for, it raises by name."""
    get = (lambda k, d=None: (cfg.get(k, d) if cfg is not None else d))
    da = get("depth_augmentations", {}) or {}
    if train and da.get("add_depth_noise", False):
        raise NotImplementedError("depth_augmentations.add_depth_noise uses open3d's DepthNoiseSimulator on the host "
                                  "(data/depth_augmentations.py:20-32); not part of this build")
    shift = None
    if train and da.get("random_depth_shift", False):
        lo, hi = float(da.get("min_shift", -0.2)), float(da.get("max_shift", 0.2))
        shift = torch.tensor([rng.uniform(lo, hi) for _ in range(depth.shape[0])], dtype=torch.float32,
                             device=depth.device)
    use_mask = mask if get("mask_depth", True) else None
    return ops.depth_frames(depth.float(), use_mask.float() if use_mask is not None else None, shift,
                            standardize=bool(get("standardize_depth", False)))

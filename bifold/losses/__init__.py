"""Loss plugin surface: mirrors `bifold.losses` (reference bifold/losses/__init__.py:5-155).

`Losses.get_by_name(cfg)` -> callable(output, sample) -> (loss, dict of per-head losses).
The hot-path loss (conf/loss/bce_gaussmap.yaml) runs on the HIP BCE kernels; its backward
hands d(loss)/d(heatmap) to the model's fused tail, exactly where autograd would.
"""
from __future__ import annotations

import torch

from bifold_amd.hip import ops


def _pop_name(cfg):
    name = cfg["name"] if isinstance(cfg, dict) else cfg.name
    try:
        del cfg["name"]
    except TypeError:
        del cfg.name
    return name


class _BCEMeanFn(torch.autograd.Function):
    """nn.BCELoss(reduction="mean") (losses/__init__.py:61) including PyTorch's clamp of each
    log at -100 and its backward's 1e-12 floor on p(1-p)."""

    @staticmethod
    def forward(ctx, heat, target):
        heat = heat.contiguous()
        target = target.contiguous()
        ctx.save_for_backward(heat, target)
        return ops.bce_mean(heat, target).view(())

    @staticmethod
    def backward(ctx, g):
        heat, target = ctx.saved_tensors
        gs = g.reshape(1).to(torch.float32).contiguous()
        return ops.bce_mean_bwd(heat, target, gs), None


def bce_mean(heat: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    if target.requires_grad:
        raise RuntimeError("BCE targets must not require grad")
    return _BCEMeanFn.apply(heat, target)


class BCEGaussMap:
    """Sum over heads of the mean BCE between predicted and Gaussian target heatmaps
    (losses/__init__.py:54-109)."""

    def __init__(self, is_bimanual, mask_pick_heatmap, **kwargs):
        self.is_bimanual = is_bimanual
        self.mask_pick_heatmap = mask_pick_heatmap

    def heads(self):
        if self.is_bimanual:
            return [f"{arm}_{act}" for arm in ("left", "right") for act in ("pick", "place")]
        return ["pick", "place"]

    def __call__(self, output, sample):
        parts = {}
        total = None
        for name in self.heads():
            target = sample[f"{name}_heatmap"]
            if self.mask_pick_heatmap and name.endswith("pick"):
                target = target * sample["mask"].squeeze(1)
            cur = bce_mean(output[f"{name}_heatmap"], target)
            parts[name] = cur
            total = cur if total is None else total + cur
        return total, parts


class BCEMask:
    """losses/__init__.py:112-117 (needs a model built with compute_mask=True)."""

    def __call__(self, output, sample):
        return bce_mean(output["mask_heatmap"], sample["mask"].squeeze(1).contiguous()), {}

    def __init__(self, **kwargs):
        pass


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target):
        prob = prob.contiguous()
        loss, stats = ops.dice_fwd(prob, target)
        ctx.save_for_backward(prob, target, stats)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        prob, target, stats = ctx.saved_tensors
        return ops.dice_bwd(prob, target, stats, g.reshape(1).to(torch.float32).contiguous()), None


class DiceLoss:
    """losses/__init__.py:126-136 on the mask head: per sample 1 - (2 sum(p t) + 1) / (sum p + sum t + 1), summed
    over the batch (bf_dice_fwd / _bwd)."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, output, sample):
        target = sample["mask"].flatten(1).contiguous()
        if target.requires_grad:
            raise RuntimeError("dice targets must not require grad")
        return _DiceFn.apply(output["mask_heatmap"].flatten(1), target), {}


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target, alpha, gamma):
        prob = prob.contiguous()
        ctx.save_for_backward(prob, target)
        ctx.alpha, ctx.gamma = alpha, gamma
        return ops.focal_fwd(prob, target, alpha, gamma).view(())

    @staticmethod
    def backward(ctx, g):
        prob, target = ctx.saved_tensors
        gs = g.reshape(1).to(torch.float32).contiguous()
        return ops.focal_bwd(prob, target, gs, ctx.alpha, ctx.gamma), None, None, None


class FocalLoss:
    """losses/__init__.py:139-155 on the mask head: alpha_t * BCE * (1 - p_t)^gamma, `.mean(1).sum()`
    (bf_focal_fwd / _bwd)."""

    def __init__(self, alpha=0.25, gamma=2, **kwargs):
        self.alpha = alpha
        self.gamma = gamma

    def __call__(self, output, sample):
        target = sample["mask"].squeeze(1).contiguous()
        if target.requires_grad:
            raise RuntimeError("focal targets must not require grad")
        return _FocalFn.apply(output["mask_heatmap"], target, float(self.alpha), float(self.gamma)), {}


class ComposedLoss:
    """Weighted sum of named losses (losses/__init__.py:30-51)."""

    def __init__(self, loss_names, weights, **kwargs):
        assert len(loss_names) == len(weights)
        self.loss_dict = {n: Losses.get_by_name(name=n, **kwargs) for n in loss_names}
        self.weights = dict(zip(loss_names, weights))

    def __call__(self, output, sample):
        total, parts = None, {}
        for n, fn in self.loss_dict.items():
            cur, sub = fn(output, sample)
            total = cur * self.weights[n] if total is None else total + cur * self.weights[n]
            parts[n] = cur
            for k, v in sub.items():
                parts[f"{n} {k}"] = v
        return total, parts


_LOSSES = {"bce_gaussmap": BCEGaussMap, "bce_mask": BCEMask, "composed": ComposedLoss, "dice": DiceLoss,
           "focal": FocalLoss}


class Losses:
    @staticmethod
    def get_by_name(cfg=None, name=None, *args, **kwargs):
        if cfg is not None:
            name = cfg["name"] if isinstance(cfg, dict) else cfg.name
        if name not in _LOSSES:
            raise ValueError(f"Loss {name} not recognized")
        if cfg is not None:
            _pop_name(cfg)
            return _LOSSES[name](*args, **cfg, **kwargs)
        return _LOSSES[name](*args, **kwargs)

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
        return bce_mean(output["mask_heatmap"], sample["mask"].squeeze(1)), {}

    def __init__(self, **kwargs):
        pass


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


_LOSSES = {"bce_gaussmap": BCEGaussMap, "bce_mask": BCEMask, "composed": ComposedLoss}
# dice / focal act on the optional mask head (losses/__init__.py:126-155), which no shipped
# model config enables (compute_mask=False): outside this build's hot path.
_NOT_BUILT = {"dice", "focal"}


class Losses:
    @staticmethod
    def get_by_name(cfg=None, name=None, *args, **kwargs):
        if cfg is not None:
            name = cfg["name"] if isinstance(cfg, dict) else cfg.name
        if name in _NOT_BUILT:
            raise NotImplementedError(f"Loss {name} is outside this build's hot path (mask head only)")
        if name not in _LOSSES:
            raise ValueError(f"Loss {name} not recognized")
        if cfg is not None:
            _pop_name(cfg)
            return _LOSSES[name](*args, **cfg, **kwargs)
        return _LOSSES[name](*args, **kwargs)

"""Static loss scaling for the fp16 compute mode (`model.dtype: fp16`).

The reference trains in fp32 (bifold/__main__.py:135-150: forward, loss, backward, clip, step) and has nothing like
this; it exists because the build's 16-bit gradient tensors (dY operands of the dX / dW GEMMs, dQKV, dO) would
underflow in IEEE binary16: `nn.BCELoss`'s mean puts 1 / (B S^2) in front of every logit gradient (8e-8 at B = 256,
S = 224; binary16's smallest subnormal is 6e-8).  bf16 has fp32's exponent and needs none of it.

Scheme: the gradient ENTERING the trainable tail's backward (the low-resolution logit gradients of the decoders,
fp32) is multiplied by S, so every tensor of the fused backward -- and every parameter gradient it writes into the
flat arena -- carries a factor S; each arena slice is multiplied by 1 / S as the backward completes it, BEFORE the
slice is reported to the gradient-ready hooks (data-parallel all-reduce, clipping, the optimiser), so `.grad` never
shows a scaled value.  S is a power of two (no rounding of its own); by default 64 x (elements of one head's
heatmaps) rounded up: measured on config A's gradients that puts the largest 16-bit gradient element at ~1e2 and the
1 % quantile at ~6e-3 (binary16: max 65 504, smallest normal 6e-5).  `loss_scale: <number>` overrides it.  Non-finite
results of the un-scaling are counted on the device (`nonfinite`, read by `overflowed()`).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch

from bifold_amd.hip import ops


class StaticGradScale:
    def __init__(self, model, setting="auto"):
        self.model = model
        if setting not in (None, "auto"):
            setting = float(setting)
            if not (setting > 0 and math.isfinite(setting)):
                raise ValueError(f"loss_scale must be a positive number or 'auto', got {setting}")
        self.setting = None if setting in (None, "auto") else setting
        self.scale = 1.0
        self.nonfinite: Optional[torch.Tensor] = None
        self._buckets: Optional[List[Tuple[int, int]]] = None
        self._arena = None
        self._depth = 0

    def _plan(self):
        from bifold.dist import layer_buckets
        model = self.model
        arena = model.trainable_arena()
        if self._buckets is None or self._arena is not arena:
            named = model.trainable_params()
            offsets = {n: arena.offsets[arena.name_of(p)] for n, p in named}
            self._depth = model.pick_place.fusion.depth
            self._buckets = layer_buckets([n for n, _ in named], offsets, self._depth, arena.count)
            self._arena = arena
            self.nonfinite = torch.zeros(1, dtype=torch.int32, device=arena.grad.device)
        return arena

    def begin(self, d_heats) -> float:
        """Called at the top of the tail's backward; returns S for this step."""
        self._plan()
        if self.setting is not None:
            self.scale = self.setting
        else:
            n = max((t.numel() for t in d_heats if t is not None), default=1)
            self.scale = 2.0 ** (math.ceil(math.log2(max(n, 1))) + 6)
        return self.scale

    def ready(self, layer: int) -> None:
        """The backward's `on_layer_done`: un-scale what layer `layer` completed, then tell the model's hooks."""
        arena = self._plan()
        depth = self._depth
        if layer == -1:
            which = range(depth + 1, len(self._buckets))
        elif layer == depth - 1:
            which = (0, 1)
        else:
            which = (depth - layer,)
        inv = 1.0 / self.scale
        for j in which:
            a, b = self._buckets[j]
            if b > a:
                ops.scale_check_(arena.grad[a:b], inv, self.nonfinite)
        self.model._grad_ready(layer)

    def overflowed(self) -> int:
        """Non-finite gradient elements seen since the last call (synchronises; diagnostics and tests)."""
        if self.nonfinite is None:
            return 0
        n = int(self.nonfinite.item())
        if n:
            self.nonfinite.zero_()
        return n


def scaler_of(model) -> Optional[StaticGradScale]:
    """The model's scaler when its compute dtype needs one (float16), else None."""
    if getattr(model, "compute_dtype", None) != torch.float16:
        return None
    gs = getattr(model, "_grad_scaler", None)
    if gs is None:
        gs = StaticGradScale(model, getattr(model, "loss_scale", "auto"))
        model._grad_scaler = gs
    return gs

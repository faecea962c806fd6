"""Loss scaling for the fp16 compute mode (`model.dtype: fp16`), with a step that survives an overflow.

The reference trains in fp32 (bifold/__main__.py:135-152: forward, loss, backward, clip, step) and has nothing like
this; it exists because the build's 16-bit gradient tensors (dY operands of the dX / dW GEMMs, dQKV, dO) would
underflow in IEEE binary16: `nn.BCELoss`'s mean puts 1 / (B S^2) in front of every logit gradient (8e-8 at B = 256,
S = 224; binary16's smallest subnormal is 6e-8).  bf16 has fp32's exponent and needs none of it.

Scheme: the gradient ENTERING the trainable tail's backward (the low-resolution logit gradients of the decoders,
fp32) is multiplied by S, so every tensor of the fused backward -- and every parameter gradient it writes into the
flat arena -- carries a factor S; each arena slice is multiplied by 1 / S as the backward completes it, BEFORE the
slice is reported to the gradient-ready hooks (data-parallel all-reduce, clipping, the optimiser), so `.grad` never
shows a scaled value to them (with no hook registered -- one process -- the whole arena is un-scaled in one pass at
the end of the backward).  S is a power of two (no rounding of its own); it starts at 64 x (elements of one head's
heatmaps) rounded up: measured on config A's gradients that puts the largest 16-bit gradient element at ~1e2 and the
1 % quantile at ~6e-3 (binary16: max 65 504, smallest normal 6e-5) -- or lower, where the first backward's largest logit
gradient says so (sum-type / weighted losses on the mask head: dice, focal, composed).  `loss_scale: <number>` overrides the start.

Round 4: S and the step's fate live ON THE DEVICE (`state`, `nonfinite`), nothing is read back per step:
  * bf_scale_check counts the non-finite results of the un-scaling in `nonfinite` (under data parallelism the
    counter is MAX-reduced over the ranks with the last gradient bucket, bifold/dist.py, so every rank decides alike);
  * bf_adam_step is handed the counter: non-zero -> parameters, moments and the 16-bit copy stay as they were (the
    step is dropped, exactly what torch.cuda.amp.GradScaler does, and it does not count for the bias corrections);
  * bf_gradscale_update then halves S after a dropped step, doubles it back after `growth_interval` clean steps
    (never past the starting value: no periodic probing, so a clean run drops nothing), and clears the counter.
The Trainer reads the totals once per epoch, reports dropped steps and raises only when S has collapsed to its floor.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from bifold_amd.hip import ops

# state slots of bf_gradscale_update (include/bifold_hip.h)
_S, _INV, _CLEAN, _SKIPPED, _BAD, _AT_FLOOR = range(6)


class DynamicGradScale:
    backoff = 0.5
    growth = 2.0
    growth_interval = 1000
    min_scale = 1.0

    def __init__(self, model, setting="auto"):
        self.model = model
        if setting not in (None, "auto"):
            setting = float(setting)
            if not (setting > 0 and math.isfinite(setting)):
                raise ValueError(f"loss_scale must be a positive number or 'auto', got {setting}")
        self.setting = None if setting in (None, "auto") else setting
        self.start_scale: Optional[float] = None
        self.state: Optional[torch.Tensor] = None          # fp32 [8] on the device
        self.nonfinite: Optional[torch.Tensor] = None      # int32 [1]: this step's non-finite gradient elements
        self.skipped_known = 0.0                           # host's last reading of state[_SKIPPED]
        self._reported = (0.0, 0.0)
        self._buckets: Optional[List[Tuple[int, int]]] = None
        self._arena = None
        self._depth = 0

    # ------------------------------------------------------------------ plumbing
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
            dev = arena.grad.device
            if self.nonfinite is None or self.nonfinite.device != dev:
                self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)
                if self.state is not None:
                    self.state = self.state.to(dev)
            arena.grad_scaler = self                       # where FusedAdam.step finds the guard
        return arena

    @staticmethod
    def _peak_logit_gradient(d_heats, saved) -> Optional[float]:
        """max |d loss / d logit| over every decoder of the first backward (the quantity the scale multiplies): for a plain
        head d_heat * p (1 - p); with the mask head (models/pickplace.py:76-98) the pick heads' d_out * mask * s (1 - s) and
        the mask's own (d_mask + sum d_out * s) * m (1 - m).  Torch glue, ONE host read, first backward of a run only."""
        if not saved or saved.get("heat") is None:
            return None
        heat, extra = saved["heat"], saved.get("extra")
        nh = heat.shape[0]
        n_pick = nh // 2
        peak = heat.new_zeros(())
        d_mask = None
        for h in range(nh):
            d = d_heats[h] if h < len(d_heats) else None
            if d is None:
                continue
            s = heat[h]
            if extra is not None and h < n_pick:
                d_mask = d * s if d_mask is None else d_mask + d * s
                d = d * extra["mask"]
            peak = torch.maximum(peak, (d * s * (1.0 - s)).abs().max())
        if extra is not None:
            m = extra["mask"]
            d_in = d_heats[nh] if len(d_heats) > nh else None
            if d_mask is None or saved.get("detach_mask"):
                d_mask = d_in
            elif d_in is not None:
                d_mask = d_mask + d_in
            if d_mask is not None:
                peak = torch.maximum(peak, (d_mask * m * (1.0 - m)).abs().max())
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            t = peak.detach().reshape(1).clone()
            if torch.distributed.get_backend() == "gloo":
                t = t.cpu()
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)     # every rank starts at the same scale
            return float(t.item())
        return float(peak.item())

    def _init_state(self, d_heats, saved=None) -> None:
        if self.setting is not None:
            s0 = self.setting
        else:
            n = max((t.numel() for t in d_heats if t is not None), default=1)
            s0 = 2.0 ** (math.ceil(math.log2(max(n, 1))) + 6)
            # 64 x elements is sized for nn.BCELoss's mean (logit gradients <= 1 / elements).  A loss that is a SUM, or
            # weighted (dice / focal / composed on the mask head), has larger ones: start where the largest scaled
            # logit gradient is ~64, like the BCE case -- never above the BCE start, which stays what it was
            peak = self._peak_logit_gradient(d_heats, saved)
            if peak is not None and math.isfinite(peak) and peak > 0.0:
                s0 = max(min(s0, 2.0 ** math.ceil(math.log2(64.0 / peak))), self.min_scale)
        self.start_scale = float(s0)
        st = torch.zeros(8, dtype=torch.float32)
        st[_S], st[_INV] = s0, 1.0 / s0
        self.state = st.to(self.nonfinite.device)

    @property
    def scale(self) -> float:
        """The current S (synchronises: diagnostics, messages and tests only)."""
        return float(self.state[_S].item()) if self.state is not None else 1.0

    # ------------------------------------------------------------------ the backward's two ends
    def begin(self, d_heats, saved=None) -> torch.Tensor:
        """Called at the top of the tail's backward (`saved`: what fusion_decoder_fwd kept -- the heads' sigmoids, the
        mask head's -- for the start of the scale); returns S as a device scalar."""
        arena = self._plan()
        if self.state is None:
            # a checkpoint loaded before the scaler existed left its scale on the arena (FusedAdam.load_state_dict): the
            # FIRST resumed backward must already run at the saved scale -- applied only inside the first step(), the
            # backward ran at the default start scale, and a run that had backed off dropped that step and halved the
            # restored scale again (advisor finding, round 4)
            pending = getattr(arena, "pending_grad_scaler_sd", None)
            if pending is not None:
                self.load_state_dict(pending)
                arena.pending_grad_scaler_sd = None
            else:
                self._init_state(d_heats, saved)
        return self.state[_S:_S + 1]

    def ready(self, layer: int) -> None:
        """The backward's `on_layer_done`: un-scale what layer `layer` completed, then tell the model's hooks."""
        arena = self._plan()
        if not getattr(self.model, "_grad_hooks", None):
            # nobody waits for single layers (one process: no gradient-ready hook is registered): ONE pass over the whole
            # arena when the backward is complete instead of eleven small launches between the layers of the chain
            if layer == -1:
                ops.scale_check_(arena.grad, self.state[_INV:_INV + 1], self.nonfinite)
            self.model._grad_ready(layer)
            return
        depth = self._depth
        if layer == -1:
            which = range(depth + 1, len(self._buckets))
        elif layer == depth - 1:
            which = (0, 1)
        else:
            which = (depth - layer,)
        inv = self.state[_INV:_INV + 1]
        for j in which:
            a, b = self._buckets[j]
            if b > a:
                ops.scale_check_(arena.grad[a:b], inv, self.nonfinite)
        self.model._grad_ready(layer)

    # ------------------------------------------------------------------ the optimiser's two ends
    def guard(self):
        """(skip flag, dropped-steps counter, the part of it the host already knows) for bf_adam_step."""
        return self.nonfinite, self.state[_SKIPPED:_SKIPPED + 1], self.skipped_known

    def update(self) -> None:
        """After the guarded step: the device-side schedule of S; clears the flag."""
        ops.gradscale_update_(self.state, self.nonfinite, self.backoff, self.growth, self.growth_interval,
                              self.min_scale, self.start_scale)

    # ------------------------------------------------------------------ host readings (synchronise; once per epoch)
    def read(self) -> dict:
        if self.state is None:
            return dict(scale=1.0, skipped=0, nonfinite=0, at_floor=0)
        st = self.state.tolist()
        self.skipped_known = st[_SKIPPED]
        return dict(scale=st[_S], skipped=int(st[_SKIPPED]), nonfinite=int(st[_BAD]), at_floor=int(st[_AT_FLOOR]))

    def overflowed(self) -> int:
        """Steps dropped since the last call (synchronises; diagnostics and tests)."""
        r = self.read()
        new = r["skipped"] - self._reported[0]
        self._reported = (r["skipped"], r["at_floor"])
        return int(new)

    # checkpoints: the scale and the counters travel with the optimiser state (FusedAdam.state_dict)
    def state_dict(self) -> dict:
        r = self.read()
        return dict(scale=r["scale"], start_scale=self.start_scale, skipped=r["skipped"],
                    clean=float(self.state[_CLEAN].item()) if self.state is not None else 0.0)

    def load_state_dict(self, sd: dict) -> None:
        self._plan()
        self.start_scale = float(sd.get("start_scale") or sd["scale"])
        st = torch.zeros(8, dtype=torch.float32)
        st[_S], st[_INV], st[_CLEAN] = float(sd["scale"]), 1.0 / float(sd["scale"]), float(sd.get("clean", 0.0))
        self.state = st.to(self.nonfinite.device)          # dropped steps are already out of the saved step count
        self.skipped_known = 0.0
        self._reported = (0.0, 0.0)


StaticGradScale = DynamicGradScale      # the name rounds 3's call sites and tests import


def scaler_of(model) -> Optional[DynamicGradScale]:
    """The model's scaler when its compute dtype needs one (float16), else None."""
    if getattr(model, "compute_dtype", None) != torch.float16:
        return None
    gs = getattr(model, "_grad_scaler", None)
    if gs is None:
        gs = DynamicGradScale(model, getattr(model, "loss_scale", "auto"))
        model._grad_scaler = gs
    return gs

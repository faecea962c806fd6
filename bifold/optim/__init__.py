"""Optimiser / scheduler plugin surface: mirrors `bifold.optim`
(reference bifold/optim/__init__.py:4-92).

`Optimizers.get_by_name(cfg, params)` returns a FusedAdam: torch.optim.Adam / AdamW math as
ONE HIP launch over the flat parameter arena (bf_adam_step), exposing the attributes the
Trainer touches (`zero_grad`, `step`, `param_groups`, `state_dict`, `load_state_dict`;
bifold/__main__.py:115,139,150,158,294,319).  `Schedulers.get_by_name` returns None or a
LinearWarmup that reproduces the reference's LR sequence including its first-step quirk.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import os

import torch

from bifold_amd.hip import ops
from ..models.arena import ParamArena, arena_of


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _pop_name(cfg):
    name = cfg["name"] if isinstance(cfg, dict) else cfg.name
    try:
        del cfg["name"]
    except TypeError:
        del cfg.name
    return name


class FusedAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, decoupled=False, **unused):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        if any(not p.is_cuda for p in self.params):
            raise RuntimeError("FusedAdam updates parameters in HBM: move the model to the GPU first")
        arena = arena_of(self.params)
        if arena is None:
            arena = ParamArena([(f"p{i}", p) for i, p in enumerate(self.params)], with_grad=True,
                               with_shadow=True)
        self.arena = arena
        self.decoupled = bool(decoupled)
        self.param_groups = [dict(params=self.params, lr=float(lr), betas=tuple(float(b) for b in betas),
                                  eps=float(eps), weight_decay=float(weight_decay), amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False,
                                  fused=True)]
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.step_count = 0
        self.grad_scale: Optional[torch.Tensor] = None       # device scalar: clip coef and/or 1/world
        self._scaler_sd = None                               # fp16 scaler state of a loaded checkpoint, until the scaler exists

    # -- Trainer-facing API ---------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False) -> None:
        """Gradients are OVERWRITTEN by every backward pass of the fused tail (they are not
        accumulated across backward calls), so there is nothing to clear; the call marks the
        arena's gradients as consumed (a second backward without it or step() raises)."""
        self.grad_scale = None
        self.arena.grads_pending = False

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ (bifold/__main__.py:148-149) on the flat gradient arena:
        three launches, no host sync; scales `.grad` in place and returns the total norm like torch.
        `torch.nn.utils.clip_grad_norm_(model.parameters(), c)` itself lands here (see below)."""
        return ops.clip_grad_norm_(self.arena.grad, float(max_norm))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        g = self.param_groups[0]
        self.step_count += 1
        # fp16 mode (models/gradscale.py): the step is GUARDED by the device-side overflow flag of this backward --
        # non-zero leaves parameters, moments and the 16-bit copy untouched, and a dropped step does not advance
        # Adam's t: `step_count` counts calls, the device counts drops, the kernel takes the difference.
        gs = getattr(self.arena, "grad_scaler", None)
        if gs is not None and gs.state is not None and self._scaler_sd is not None:
            if getattr(self.arena, "pending_grad_scaler_sd", None) is not None:
                gs.load_state_dict(self._scaler_sd)        # (a scaler whose state was built without passing begin())
                self.arena.pending_grad_scaler_sd = None
            self._scaler_sd = None
        flag, skipped, known = gs.guard() if gs is not None and gs.state is not None else (None, None, 0.0)
        ops.adam_step(self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, self.arena.shadow,
                      self.step_count - int(known), g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                      self.decoupled, self.grad_scale, flag, skipped, known)
        if flag is not None:
            gs.update()
        self.arena.mark_synced()
        self.arena.grads_pending = False

    def _applied_steps(self) -> int:
        """Optimiser steps that changed the parameters (calls minus the steps the fp16 guard dropped)."""
        gs = getattr(self.arena, "grad_scaler", None)
        if gs is None or gs.state is None:
            return self.step_count
        return self.step_count - gs.read()["skipped"]

    # -- checkpoint format of torch.optim.Adam --------------------------------------------
    def state_dict(self) -> Dict:
        state = {}
        applied = self._applied_steps()
        if applied > 0:
            for i, p in enumerate(self.params):
                o, cnt = self.arena.offsets[self.arena.name_of(p)]
                state[i] = {"step": torch.tensor(float(applied)),
                            "exp_avg": self.exp_avg[o:o + cnt].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + cnt].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        out = {"state": state, "param_groups": [group]}
        gs = getattr(self.arena, "grad_scaler", None)
        if gs is not None and gs.state is not None:       # extra key: torch.optim.Adam.load_state_dict ignores it
            out["bifold_grad_scaler"] = gs.state_dict()
        return out

    def load_state_dict(self, sd: Dict) -> None:
        group = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in group:
                self.param_groups[0][k] = tuple(group[k]) if k == "betas" else group[k]
        steps = set()
        for i, st in sd.get("state", {}).items():
            p = self.params[int(i)]
            o, cnt = self.arena.offsets[self.arena.name_of(p)]
            self.exp_avg[o:o + cnt].view(p.shape).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + cnt].view(p.shape).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; FusedAdam keeps one step counter")
        self.step_count = steps.pop() if steps else 0
        gs = getattr(self.arena, "grad_scaler", None)
        self._scaler_sd = sd.get("bifold_grad_scaler")
        # (no scaler yet: it is built by the first backward, which takes the saved scale from the arena BEFORE it runs --
        # DynamicGradScale.begin; the copy kept here only covers a scaler that exists without a state)
        self.arena.pending_grad_scaler_sd = self._scaler_sd if (gs is None or gs.state is None) else None
        if gs is not None and gs.state is not None:
            if self._scaler_sd is not None:
                gs.load_state_dict(self._scaler_sd)
                self._scaler_sd = None
            else:
                gs.skipped_known = 0.0                     # the loaded step count is net of drops
                gs.state[3] = 0.0


# ---- torch.nn.utils.clip_grad_norm_(model.parameters(), c), unchanged at the call site -----------
# The reference trainer clips with the torch utility (bifold/__main__.py:148-149).  On arena-backed
# gradients that call would still work -- it sees ~110 `.grad` views -- but as a per-tensor foreach norm,
# a stack, a second norm and a foreach multiply.  Importing this module wraps the utility: when every
# gradient it is handed is a view of ONE ParamArena (and the call asks for the plain L2 norm), the three
# launches of bf_clip_grad_norm run over the flat buffer instead; anything else goes to torch's own
# implementation untouched.
_torch_clip_grad_norm_ = getattr(torch.nn.utils.clip_grad_norm_, "_bifold_original", torch.nn.utils.clip_grad_norm_)


def _arena_behind(params: List[torch.Tensor]) -> Optional[ParamArena]:
    """The ParamArena whose gradient buffer backs `.grad` of EVERY parameter that has one, provided
    those parameters are all of the arena's; None otherwise."""
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return None
    arena = getattr(with_grad[0], "_bf_arena", None)
    if arena is None or arena.grad is None or len(with_grad) != len(arena.params):
        return None
    base, ids = arena.grad.data_ptr(), {id(p) for p in arena.params}
    for p in with_grad:
        if id(p) not in ids or p.grad.data_ptr() != base + 4 * arena.offsets[arena.name_of(p)][0]:
            return None
    return arena


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = list(parameters)
    if float(norm_type) == 2.0 and not error_if_nonfinite:
        arena = _arena_behind(params)
        if arena is not None:
            return ops.clip_grad_norm_(arena.grad, float(max_norm))
    return _torch_clip_grad_norm_(params, max_norm, norm_type=norm_type, error_if_nonfinite=error_if_nonfinite,
                                  foreach=foreach)


clip_grad_norm_._bifold_original = _torch_clip_grad_norm_


def wrap_torch_clip_grad_norm(enable: bool = True) -> None:
    """Install (or remove) the arena-aware `clip_grad_norm_` as `torch.nn.utils.clip_grad_norm_`.  The wrap is process-wide
    state a host application may not expect from an import; it exists so that the reference trainer's clipping line runs
    unedited.  Opt out with `BIFOLD_NO_CLIP_WRAP=1` in the environment, or call `wrap_torch_clip_grad_norm(False)` at any
    time: torch's own function is put back, and `bifold.optim.clip_grad_norm_(params, c)` / `optimizer.clip_grad_norm_(c)`
    remain available as explicit calls."""
    fn = clip_grad_norm_ if enable else _torch_clip_grad_norm_
    torch.nn.utils.clip_grad_norm_ = fn
    if getattr(torch.nn.utils, "clip_grad", None) is not None:
        torch.nn.utils.clip_grad.clip_grad_norm_ = fn


if os.environ.get("BIFOLD_NO_CLIP_WRAP") != "1":
    wrap_torch_clip_grad_norm(True)


class Optimizers:
    @staticmethod
    def get_by_name(cfg, params):
        name = _cfg_get(cfg, "name")
        if name not in ("adam", "adamw"):
            raise ValueError(f"Optimizer {name} not recognized")
        _pop_name(cfg)
        return FusedAdam(params, decoupled=(name == "adamw"), **cfg)


class LinearWarmup:
    """LR = start + (base - start) * last_step / warmup_steps while last_step <= warmup_steps,
    then cosine annealing over max_iters - warmup_steps steps (optim/__init__.py:28-92).
    Quirk kept on purpose: `last_step` starts at -1 and `step()` already runs in __init__, so
    the first optimiser step uses a NEGATIVE learning rate when warmup_start_lr = 0."""

    def __init__(self, cfg, optimizer, max_iters):
        self.warmup_steps = int(_cfg_get(cfg, "warmup_portion") * max_iters)
        self.optimizer = optimizer
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_step = -1
        start = _cfg_get(cfg, "warmup_start_lr")
        if isinstance(start, (list, tuple)):
            assert len(start) == len(self.base_lrs), (
                f"The length of warmup_start_lr {len(start)} and optimizer.param_group "
                f"{len(self.base_lrs)} do not correspond")
            self.warmup_start_lrs = list(start)
        else:
            self.warmup_start_lrs = [start] * len(self.base_lrs)
        self.use_cosine_decay = bool(_cfg_get(cfg, "use_cosine_decay"))
        self.t_max = max_iters - self.warmup_steps
        self.cosine_step = 0
        self.step()

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def step(self, epoch=None):
        if self.last_step <= self.warmup_steps:
            for g, start, base in zip(self.optimizer.param_groups, self.warmup_start_lrs, self.base_lrs):
                g["lr"] = start + (base - start) * (self.last_step / self.warmup_steps)
            self.last_step += 1
        elif self.use_cosine_decay:
            # closed form of torch's CosineAnnealingLR (eta_min = 0) started from base_lr
            self.cosine_step += 1
            for g, base in zip(self.optimizer.param_groups, self.base_lrs):
                g["lr"] = base * (1.0 + math.cos(math.pi * self.cosine_step / self.t_max)) / 2.0
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]


class Schedulers:
    @staticmethod
    def get_by_name(cfg, *args, **kwargs):
        name = _cfg_get(cfg, "name")
        if name is None:
            return None
        if name == "linear_warmup":
            return LinearWarmup(cfg, *args, **kwargs)
        raise ValueError(f"Scheduler {name} not recognized")

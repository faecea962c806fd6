"""Flat parameter arenas.

All trainable parameters of a model live in ONE contiguous fp32 buffer (and their
gradients in a second one, their 16-bit (bf16 / fp16) GEMM-operand copies in a third).  nn.Parameters
stay ordinary Parameters -- `state_dict()`, `named_parameters()`, checkpoints keep the
reference's key names -- but their `.data` / `.grad` are views into the arenas, so that

  * the optimiser is ONE fused launch over the arena (bf_adam_step),
  * gradient all-reduce works on contiguous slices (no flatten/unflatten copies),
  * the bf16 operand copies are refreshed by the optimiser kernel itself.

Sized for MI355X: config A's 26 M trainable parameters are 105 MB per arena out of 288 GB.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
from torch import nn

ALIGN = 64  # elements; keeps every parameter 256-byte aligned (GEMM operands need 16 B)


class ParamArena:
    def __init__(self, named_params: Iterable[Tuple[str, nn.Parameter]], *, with_grad: bool,
                 with_shadow: bool, shadow_dtype: torch.dtype = torch.bfloat16):
        items = [(n, p) for n, p in named_params]
        if not items:
            raise ValueError("ParamArena needs at least one parameter")
        dev = items[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("ParamArena lives in HBM: move the model to the GPU first")
        self.names: List[str] = []
        self.params: List[nn.Parameter] = []
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n, p in items:
            if p.dtype != torch.float32:
                raise TypeError(f"parameter {n} must be float32 (master copy), got {p.dtype}")
            self.offsets[n] = (off, p.numel())
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.names.append(n)
            self.params.append(p)
        self.count = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev) if with_grad else None
        self.shadow = torch.zeros(off, dtype=shadow_dtype, device=dev) if with_shadow else None
        for n, p in items:
            o, cnt = self.offsets[n]
            view = self.flat[o:o + cnt].view(p.shape)
            view.copy_(p.data)
            p.data = view
            if with_grad and p.requires_grad:
                p.grad = self.grad[o:o + cnt].view(p.shape)
            p._bf_arena = self
            p._bf_name = n
        self._by_id = {id(p): n for n, p in items}
        self._versions = -1
        self.grads_pending = False        # a backward wrote gradients that no step()/zero_grad() consumed yet
        self.sync_shadow(force=True)

    # -- views -----------------------------------------------------------------
    def name_of(self, p: nn.Parameter) -> str:
        return self._by_id[id(p)]

    def master(self, name: str) -> torch.Tensor:
        o, cnt = self.offsets[name]
        return self.flat[o:o + cnt]

    def grad_view(self, p: nn.Parameter) -> torch.Tensor:
        """The arena slice that receives this parameter's gradient; re-attached as
        `p.grad` if something (e.g. zero_grad(set_to_none=True)) detached it."""
        o, cnt = self.offsets[self._by_id[id(p)]]
        g = self.grad[o:o + cnt].view(p.shape)
        if p.grad is None or p.grad.data_ptr() != g.data_ptr():
            p.grad = g
        return g

    def operand(self, p: nn.Parameter, cdt: torch.dtype) -> torch.Tensor:
        """The parameter in the compute dtype (16-bit shadow view, or the fp32 master).  The shadow follows the
        compute dtype that asks for it: an arena created with bf16 copies (the optimiser may build it before the model
        runs) is re-cast once when an fp16 model takes its first operand."""
        if cdt == torch.float32:
            return p.data
        if self.shadow is None or self.shadow.dtype != cdt:
            self.set_shadow_dtype(cdt)
        o, cnt = self.offsets[self._by_id[id(p)]]
        return self.shadow[o:o + cnt].view(p.shape)

    def set_shadow_dtype(self, dtype: torch.dtype) -> None:
        if dtype not in (torch.bfloat16, torch.float16):
            raise TypeError(f"operand copies are bfloat16 or float16, got {dtype}")
        if self.shadow is None or self.shadow.dtype != dtype:
            self.shadow = torch.zeros(self.count, dtype=dtype, device=self.flat.device)
            self.sync_shadow(force=True)

    def begin_backward(self) -> None:
        """The fused backward OVERWRITES the gradient arena; a second backward before the optimiser
        consumed the first (gradient accumulation, two losses on one model) would silently drop
        gradients, so it raises instead.  Clearing `.grad` (`model.zero_grad()`) also counts as consumed."""
        if self.grads_pending and all(p.grad is not None for p in self.params if p.requires_grad):
            raise RuntimeError(
                "backward() was called twice without optimizer.step() / zero_grad() in between: the fused "
                "backward overwrites the gradient arena, gradient accumulation is not supported on the HIP path")
        self.grads_pending = True

    # -- consistency -----------------------------------------------------------
    def still_bound(self) -> bool:
        base = self.flat.data_ptr()
        for n, p in zip(self.names, self.params):
            if p.data.data_ptr() != base + 4 * self.offsets[n][0]:
                return False
        return True

    def sync_shadow(self, force: bool = False) -> None:
        """Re-cast the bf16 copies when a parameter was written outside the fused optimiser
        (load_state_dict, manual init): detected through autograd's version counters."""
        if self.shadow is None:
            return
        ver = sum(p._version for p in self.params)
        if force or ver != self._versions:
            from bifold_amd.hip import ops
            ops.cast(self.flat, self.shadow.dtype, out=self.shadow)
            self._versions = ver

    def mark_synced(self) -> None:
        self._versions = sum(p._version for p in self.params)


def arena_of(params: List[nn.Parameter]) -> Optional[ParamArena]:
    """The arena shared by exactly these parameters, if any."""
    a = getattr(params[0], "_bf_arena", None) if params else None
    if a is None or len(a.params) != len(params):
        return None
    ids = {id(p) for p in a.params}
    if any(id(p) not in ids for p in params) or not a.still_bound():
        return None
    return a

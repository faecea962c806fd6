"""Parameter trees addressed by the reference's state-dict key names.

The checkpoint schema is part of the drop-in surface, the reference's
nn.Module class hierarchy is not.  Modules here are thin containers: parameters are
registered under dotted paths (`pick_place.fusion.transformer_encoder.layers.3.0.fn.to_qkv.weight`)
by creating anonymous child modules on the way, which reproduces the key names exactly.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
from torch import nn


def add_param(root: nn.Module, path: str, value: torch.Tensor, requires_grad: bool = True) -> nn.Parameter:
    parts = path.split(".")
    mod = root
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, nn.Module())
        mod = mod._modules[part]
    p = nn.Parameter(value, requires_grad=requires_grad)
    mod.register_parameter(parts[-1], p)
    return p


def get_param(root: nn.Module, path: str) -> nn.Parameter:
    mod = root
    parts = path.split(".")
    for part in parts[:-1]:
        mod = mod._modules[part]
    return mod._parameters[parts[-1]]


# --- initialisers matching the reference's defaults -----------------------------------
def linear_init(out_f: int, in_f: int, bias: bool = True) -> Dict[str, torch.Tensor]:
    """nn.Linear / 1x1 nn.Conv2d default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in))
    for the weight and the same bound for the bias."""
    bound = 1.0 / math.sqrt(in_f)
    out = {"weight": torch.empty(out_f, in_f).uniform_(-bound, bound)}
    if bias:
        out["bias"] = torch.empty(out_f).uniform_(-bound, bound)
    return out


def layernorm_init(width: int) -> Dict[str, torch.Tensor]:
    return {"weight": torch.ones(width), "bias": torch.zeros(width)}

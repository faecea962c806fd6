"""Reading pretrained tower weights from the files a reference user already has.

The reference downloads its encoders -- `clip.load` fetches OpenAI's TorchScript archives
(bifold/models/clip/__init__.py:117-176, falling back to `torch.load` for plain state dicts,
:135-142) and `AutoModel.from_pretrained` pulls Hugging Face safetensors
(bifold/models/siglip.py:26).  There is no network here, so the towers are filled from local
copies of exactly those files:

  * a TorchScript archive (OpenAI `ViT-B-16.pt` ...): its `state_dict()`, minus the three
    bookkeeping entries `build_model` deletes (clip/model.py:550-552);
  * a `*.safetensors` file, or a Hugging Face snapshot directory holding `model.safetensors`
    or `pytorch_model.bin`;
  * a `torch.save`d state dict, bare or nested under "state_dict" / "model" (the trainer's own
    `checkpoints/last.pth` keeps the full model under "model").

Every tensor is returned as fp32 on the CPU; `load_into` then refuses a file that does not
provide every parameter of the tower (a silently half-initialised encoder trains to nonsense).
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, Iterable

import torch
from torch import nn

_CLIP_BOOKKEEPING = ("input_resolution", "context_length", "vocab_size")


def _from_torch_file(path: str) -> Dict[str, torch.Tensor]:
    try:                                   # OpenAI's archives are TorchScript modules
        return dict(torch.jit.load(path, map_location="cpu").state_dict())
    except RuntimeError:
        pass
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    if not isinstance(obj, dict) or not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: not a state dict (expected a mapping of names to tensors)")
    return dict(obj)


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    path = os.path.expanduser(str(path))
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin"):
            if os.path.isfile(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin in this directory")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"No checkpoint found at {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        sd = _from_torch_file(path)
    return {k: v.detach().float() for k, v in sd.items() if k not in _CLIP_BOOKKEEPING}


def load_into(module: nn.Module, sd: Dict[str, torch.Tensor], what: str, prefix: str = "",
              optional: Iterable[str] = (), optional_containing: Iterable[str] = ()) -> None:
    """Copy `sd` into `module` (keys optionally under `prefix`, e.g. a tower saved inside a full
    model).  Missing parameters and shape mismatches raise -- except names starting with one of
    `optional` or containing one of `optional_containing` (LoRA adapters absent from a base-model
    file); extra entries only warn."""
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    own = module.state_dict()
    optional, inside = tuple(optional), tuple(optional_containing)
    missing = [k for k in own if k not in sd and not k.startswith(optional) and not any(t in k for t in inside)]
    if missing:
        raise KeyError(f"{what}: checkpoint lacks {len(missing)} of {len(own)} tensors, e.g. {missing[:4]}")
    bad = [(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in own if k in sd and sd[k].shape != own[k].shape]
    if bad:
        k, got, want = bad[0]
        raise ValueError(f"{what}: {len(bad)} tensors have the wrong shape, e.g. {k}: file {got}, model {want} "
                         f"(is this the checkpoint of another model size?)")
    extra = [k for k in sd if k not in own]
    if extra:
        warnings.warn(f"{what}: ignoring {len(extra)} checkpoint entries the tower does not use, e.g. {extra[:3]}")
    module.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)

"""Model plugin surface: mirrors `bifold.models` (reference bifold/models/__init__.py:12-173).

Same factory names, same config-driven construction (`cfg.name` selects the class, the
remaining keys of the config node ARE the constructor kwargs), same `BaseModel` contract
(`forward(sample) -> dict`, `get_action(sample, return_raw_output)`).  What differs is
underneath: forward/backward run on libbifold_hip.so.
"""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np
import torch
from torch import nn


class Action:
    """Pick/place pixel action container (reference bifold/env/__init__.py:1-19):
    two keyword arguments -> unimanual, four -> bimanual."""

    def __init__(self, **kwargs):
        if set(kwargs) == {"pick", "place"}:
            self.pick, self.place = kwargs["pick"], kwargs["place"]
        elif set(kwargs) == {"left_pick", "right_pick", "left_place", "right_place"}:
            self.left_pick, self.right_pick = kwargs["left_pick"], kwargs["right_pick"]
            self.left_place, self.right_place = kwargs["left_place"], kwargs["right_place"]
        else:
            raise ValueError(f"Cannot instantiate action with {kwargs}")


def _lazy(module: str, attr: str) -> Callable:
    def load():
        import importlib
        return getattr(importlib.import_module(module, __package__), attr)
    return load


_MODELS: Dict[str, Callable] = {
    "rgb_clip": _lazy(".rgb_clip", "RGBOnly"),
    "siglip": _lazy(".siglip", "SigLip"),
    "siglip_sequential": _lazy(".siglip", "SiglipSequential"),
}
_COMPONENTS: Dict[str, Callable] = {
    "pick_place_convdecoder": _lazy(".pickplace", "PickPlaceConvDecoder"),
    "concat_transformer": _lazy(".pickplace", "ConcatTransformer"),
}
# Names the reference knows (models/__init__.py:15-24,33-42) that this hot-path build does
# not cover yet; they fail loudly instead of silently running something else.
_NOT_BUILT = {"text_unet", "pick_place_transdecoder", "crossattention"}


def _pop_name(cfg):
    name = cfg["name"] if isinstance(cfg, dict) else cfg.name
    try:
        del cfg["name"]
    except TypeError:
        del cfg.name
    return name


class Models:
    @staticmethod
    def get_by_name(cfg, *args, **kwargs):
        """`cfg.name` picks the model; like the reference it is deleted from the node and the
        rest is splatted into the constructor (models/__init__.py:26-27)."""
        name = cfg["name"] if isinstance(cfg, dict) else cfg.name
        if name in _MODELS:
            _pop_name(cfg)
            return _MODELS[name]()(*args, **cfg, **kwargs)
        if name in _NOT_BUILT:
            raise NotImplementedError(
                f"Model {name} is part of the reference but outside this build's hot path (see DESIGN.md)")
        raise ValueError(f"Model {name} not recognized")


class Components:
    @staticmethod
    def get_by_name(name, *args, **kwargs):
        if name in _COMPONENTS:
            return _COMPONENTS[name]()(*args, **kwargs)
        if name in _NOT_BUILT:
            raise NotImplementedError(
                f"Component {name} is part of the reference but outside this build's hot path")
        raise ValueError(f"Model {name} not recognized")


DUMMY_PICK = -np.ones(2)


def nearest_to_mask(u: int, v: int, mask: np.ndarray) -> np.ndarray:
    """Closest (row, col) of a non-zero mask pixel to (u, v) (models/utils.py:80-83)."""
    idx = np.argwhere(mask)
    return idx[((idx - [u, v]) ** 2).sum(1).argmin()]


def sample_from_heatmap(heatmap: torch.Tensor, mask=None, return_confidence: bool = False):
    """Per-sample argmax -> (x, y); optionally snapped to the mask (models/utils.py:60-77)."""
    hm = heatmap.detach().float().cpu().numpy()
    pixels, confidences = [], []
    for i in range(hm.shape[0]):
        r, c = np.unravel_index(hm[i].argmax(), hm[i].shape)
        if mask is not None:
            m = mask[i].squeeze().cpu().numpy()
            r, c = nearest_to_mask(r, c, m)
            assert m[r, c]
        pixels.append([c, r])
        confidences.append(hm[i][r, c])
    if return_confidence:
        return np.array(pixels), np.array(confidences)
    return np.array(pixels)


class BaseModel(nn.Module):
    """Constructor contract of the reference's BaseModel (models/__init__.py:46-66), including
    the quirk that the confidence threshold is fixed at 0.01 whatever the YAML says (:61)."""

    def __init__(self, image_size, is_bimanual, device, threshold=0.01, requires_graph=False,
                 constrain_pick_mask=True, **kwargs):
        super().__init__()
        self.image_size = image_size
        self.is_bimanual = is_bimanual
        self.device = device
        self.threshold = 0.01
        self.requires_graph = requires_graph
        self.constrain_pick_mask = constrain_pick_mask
        if not self.constrain_pick_mask:
            print("This run will not constrain the pick to the mask")

    def forward(self, x):
        raise NotImplementedError

    def frozen_submodule(self, submodule_name):
        for name, p in self.named_parameters():
            if submodule_name in name:
                p.requires_grad = False

    def frozen_all(self):
        for p in self.parameters():
            p.requires_grad = False

    def get_action(self, sample, return_raw_output=False):
        """Inference post-processing (models/__init__.py:79-173), heatmaps -> pixel actions."""
        if self.requires_graph:
            raise NotImplementedError("graph-based picks (requires_graph) are outside this build")
        with torch.no_grad():
            output = self.forward(sample)
        if self.is_bimanual:
            picks, confs = [], []
            for arm in ("left", "right"):
                p, c = sample_from_heatmap(output[f"{arm}_pick_heatmap"], sample["mask"], True)
                picks.append(p)
                confs.append(c)
            pick = np.stack(picks).astype(np.float64)           # (2, B, 2)
            conf = np.stack(confs)                              # (2, B)
            b = conf.shape[-1]
            # keep an arm if it is confident, or if it is the more confident of the two
            keep = np.logical_or(conf >= self.threshold,
                                 conf.argmax(axis=0) == np.tile(np.arange(2), (b, 1)).T)
            pick[~keep] = DUMMY_PICK
            places = []
            for i, arm in enumerate(("left", "right")):
                pl = sample_from_heatmap(output[f"{arm}_place_heatmap"]).astype(np.float64)
                pl[~keep[i]] = DUMMY_PICK
                places.append(pl)
            action = Action(left_pick=pick[0], right_pick=pick[1], left_place=places[0],
                            right_place=places[1])
        else:
            mask = sample.get("mask") if self.constrain_pick_mask else None
            action = Action(pick=sample_from_heatmap(output["pick_heatmap"], mask),
                            place=sample_from_heatmap(output["place_heatmap"]))
        return (action, output) if return_raw_output else action

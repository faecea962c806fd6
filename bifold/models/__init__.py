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


def _picks_to_host(picks):
    """[(xy int32 [B,2], conf fp32 [B]) ...] on the device -> int64 / fp32 numpy arrays with ONE
    device->host copy each (the reference copies every heatmap, 200 KB per sample, instead)."""
    xy = torch.stack([p[0] for p in picks]).cpu().numpy().astype(np.int64)
    conf = torch.stack([p[1] for p in picks]).cpu().numpy()
    if (xy == -2).any():          # bf_heatmap_pick's marker for an empty mask
        raise ValueError("attempt to get argmin of an empty sequence (a sample's mask has no pixel set)")
    return xy, conf


def _device_pick(heatmap: torch.Tensor, mask=None):
    from bifold_amd.hip import ops
    if not heatmap.is_cuda:
        raise RuntimeError("sample_from_heatmap runs on the MI355X (bf_heatmap_pick); the numpy restatement "
                           "of the reference lives in oracle/ for tests")
    return ops.heatmap_pick(heatmap.detach().float(), mask)


def sample_from_heatmap(heatmap: torch.Tensor, mask=None, return_confidence: bool = False):
    """Per-sample argmax -> (x, y) pixel, optionally snapped to the nearest mask pixel
    (models/utils.py:60-83), computed on the device by bf_heatmap_pick; only the [B, 2]
    pixels and [B] confidences cross to the host."""
    xy, conf = _picks_to_host([_device_pick(heatmap, mask)])
    if return_confidence:
        return xy[0], conf[0]
    return xy[0]


def select_arms(confidences: np.ndarray, threshold: float) -> np.ndarray:
    """(2, B) pick confidences -> (2, B) keep flags (models/__init__.py:118-131): an arm acts if it
    is confident, or if it is the more confident of the two -- at least one arm always acts."""
    b = confidences.shape[-1]
    return np.logical_or(confidences >= threshold,
                         confidences.argmax(axis=0) == np.tile(np.arange(2), (b, 1)).T)


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
            mask = sample["mask"]
            names = ("left_pick", "right_pick", "left_place", "right_place")
            xy, conf = _picks_to_host([_device_pick(output[f"{n}_heatmap"], mask if "pick" in n else None)
                                       for n in names])
            keep = select_arms(conf[:2], self.threshold)            # (2, B)
            pick, place = xy[:2], xy[2:]
            pick[~keep] = DUMMY_PICK
            place[~keep] = DUMMY_PICK
            action = Action(left_pick=pick[0], right_pick=pick[1], left_place=place[0],
                            right_place=place[1])
        else:
            mask = sample.get("mask") if self.constrain_pick_mask else None
            xy, _ = _picks_to_host([_device_pick(output["pick_heatmap"], mask),
                                    _device_pick(output["place_heatmap"])])
            action = Action(pick=xy[0], place=xy[1])
        return (action, output) if return_raw_output else action

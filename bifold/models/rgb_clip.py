"""RGBOnly: frozen CLIP image+text encoders -> concat-fusion transformer -> pick/place
heatmaps (reference bifold/models/rgb_clip.py:9-73; BASELINE configs 1, 2 and 4).

Constructor kwargs are the YAML keys of conf/model/rgb_clip.yaml.  Two keys are new and
default to the fast path: `dtype` ("bf16": bf16 MFMA operands, fp32 accumulate / residual
stream / statistics; "fp32": exact fp32 MFMA everywhere -- the parity mode; "fp8": bf16 with the
frozen towers' QKV / fc1 GEMMs in OCP e4m3, per-token and per-channel scales -- BASELINE configs[4]) and
`clip_checkpoint` (a CLIP state dict on disk; the reference downloads weights instead).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
from torch import nn

from . import BaseModel, Components
from .streams import serial, side_stream
from .arena import ParamArena, arena_of
from .clip_encoder import _MODELS, load
from .params import add_param, linear_init
from .pickplace import RGBTailFn

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
           "float32": torch.float32, "fp8": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16,
           "float16": torch.float16}


class RGBOnly(BaseModel):
    def __init__(self, patch_size, text_dropout, rgb_dropout, text_encoder, pick_place_model,
                 dtype="bf16", clip_checkpoint=None, loss_scale="auto", **kwargs):
        super().__init__(**kwargs)
        for name, v in (("text_dropout", text_dropout), ("rgb_dropout", rgb_dropout)):
            if not 0.0 <= float(v) < 1.0:
                raise ValueError(f"{name} must be in [0, 1), got {v}")
        # nn.Dropout on the two modalities' embeddings (ref:bifold/models/rgb_clip.py:39-42,56,65), active in training; the
        # fused tail applies the replayable masks (RGBTailFn) and records their seeds in `last_dropout_seeds`
        self.text_dropout, self.rgb_dropout = float(text_dropout), float(rgb_dropout)
        self.last_dropout_seeds = {}
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}, got {dtype!r}")
        self.compute_dtype = _DTYPES[dtype]
        self.loss_scale = loss_scale                           # fp16 mode only (models/gradscale.py)
        self.patch_size = patch_size
        self.rgb_size = self.image_size
        assert self.rgb_size % self.patch_size == 0, "Image dimensions must be divisible by the patch size."
        num_patches = (self.rgb_size // self.patch_size) ** 2

        assert text_encoder in _MODELS
        self.clip_encoder, _ = load(text_encoder, device="cpu", checkpoint=clip_checkpoint)
        self.clip_encoder.fp8 = dtype == "fp8"             # frozen towers: e4m3 QKV / fc1 GEMMs (BASELINE configs[4])
        self.frozen_submodule("clip_encoder")
        dim = self.clip_encoder.transformer.width
        if self.clip_encoder.grid ** 2 != num_patches:
            raise ValueError("image_size / patch_size do not match the CLIP vision tower")

        for k, v in linear_init(dim, self.clip_encoder.vision_width).items():
            add_param(self, f"project.{k}", v)
        self.text_token = nn.Parameter(torch.randn(1, 1, dim))
        self.text_pos_embedding = nn.Parameter(torch.randn(1, 78, dim))      # text length + 1
        self.rgb_pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.pick_place = Components.get_by_name(pick_place_model, dim=dim, patch_size=patch_size,
                                                 num_patches=num_patches, **kwargs)
        self.pick_place.fusion.compute_dtype = self.compute_dtype
        self._train_arena: Optional[ParamArena] = None
        self._grad_hooks: List[Callable[[int], None]] = []

    # ------------------------------------------------------------------ runtime plumbing
    def trainable_params(self):
        return [(n, p) for n, p in self.named_parameters() if "clip_encoder" not in n]

    def trainable_arena(self) -> ParamArena:
        """Flat fp32 master / gradient / bf16-operand arenas of the trainable parameters,
        built on first use after the model reached the GPU (shared with FusedAdam)."""
        a = self._train_arena
        if a is None or not a.still_bound():
            named = self.trainable_params()
            a = arena_of([p for _, p in named])
            if a is None or (self.compute_dtype != torch.float32 and a.shadow is None):
                a = ParamArena(named, with_grad=True, with_shadow=True)
            self._train_arena = a
        return a

    def register_grad_ready_hook(self, fn: Callable[[int], None]) -> None:
        """fn(i) is called during backward once every gradient of fusion layer i is in the
        arena (i = depth-1 ... 0), then fn(-1) once all gradients are; the data-parallel
        wrapper uses it to overlap all-reduce with the rest of backward."""
        self._grad_hooks.append(fn)

    def _grad_ready(self, layer: int) -> None:
        for fn in self._grad_hooks:
            fn(layer)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        rgb, instr = x["rgb"], x["instruction"]
        if not rgb.is_cuda:
            raise RuntimeError("RGBOnly runs on the MI355X only: move the sample and the model to the GPU "
                               "(there is no CPU path; the CPU restatement lives in oracle/ for tests)")
        cdt = self.compute_dtype
        arena = self.trainable_arena()
        arena.sync_shadow()
        self.pick_place.collapse_async()                  # decoder collapse beside the towers (side stream)
        # The text tower's GEMMs are small (77 tokens/sample: 0.6-2.4 rounds of tiles) and leave CUs
        # idle; it runs on a side HIP stream concurrently with the vision tower and joins before the
        # fusion tail.  Both towers are frozen (no autograd state crosses the streams).
        main = torch.cuda.current_stream()
        side = side_stream(rgb.device, "text")
        self.clip_encoder._runtime(cdt)                   # arenas / bf16 copies ready before the fork
        if serial():                                      # profiling mode: both towers on the main stream
            text_feat = self.clip_encoder.encode_text_with_embeddings(instr, cdt)
            vis_feat = self.clip_encoder.encode_image_with_embeddings(rgb, cdt)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                text_feat = self.clip_encoder.encode_text_with_embeddings(instr, cdt)
            vis_feat = self.clip_encoder.encode_image_with_embeddings(rgb, cdt)
            main.wait_stream(side)
            text_feat.record_stream(main)
        params = [p for _, p in self.trainable_params()]
        heats = RGBTailFn.apply(self, text_feat, vis_feat, *params)
        out = {"attn_weights": None}
        for name, h in zip(self.pick_place.heads_out, heats):
            out[f"{name}_heatmap"] = h
        if self.pick_place.has_mask:                      # models/pickplace.py:66-69
            out["mask_heatmap"] = heats[len(self.pick_place.heads_out)]
        return out

"""SigLip / SiglipSequential: frozen SigLIP towers -> concat-fusion transformer -> pick/place
heatmaps (reference bifold/models/siglip.py:10-137; `siglip_sequential` is the reference's
default model and BASELINE config 3).

Constructor kwargs are the YAML keys of conf/model/siglip*.yaml.  `lora: false` freezes the
towers (models/siglip.py:41-43).  `lora: true` (models/siglip.py:33-40, the reference's default)
adds rank-r adapters to `target_modules` of BOTH towers and trains them: the towers then keep
their activations and `tower_backward` continues the fused tail's backward through them
(siglip_encoder.SiglipTowers).  LoRA's arithmetic lives in the third-party `peft` package, absent
from this image: that mode follows PEFT's published definition and is PARITY UNPINNED (checked
against oracle/lora_oracle.py, a torch-autograd restatement, not against the reference);
`lora_dropout` > 0 (the reference ships 0.01) runs the adapters unmerged in training with a
device-generated, replayable mask per adapted Linear (siglip_encoder.SiglipTowers); in eval mode or at
0 they are merged into the in-projection weights.  New optional keys: `dtype`,
`siglip_checkpoint`.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
from torch import nn

from . import BaseModel, Components
from .arena import ParamArena, arena_of
from .pickplace import SegmentTailFn
from .siglip_encoder import from_pretrained

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
           "fp16": torch.float16, "f16": torch.float16, "float16": torch.float16}


class SigLip(BaseModel):
    def __init__(self, patch_size, dim, lora, r, lora_alpha, lora_dropout, automodel_name, target_modules,
                 pick_place_model, dtype="bf16", siglip_checkpoint=None, loss_scale="auto", lora_recompute="none",
                 **kwargs):
        super().__init__(**kwargs)
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}, got {dtype!r}")
        self.compute_dtype = _DTYPES[dtype]
        self.loss_scale = loss_scale                           # fp16 mode only (models/gradscale.py)
        self.patch_size = patch_size
        self.lora = bool(lora)
        adapters = (dict(r=r, alpha=lora_alpha, targets=list(target_modules), adapter="siglip_adapter",
                         dropout=float(lora_dropout), recompute=lora_recompute) if lora else None)
        self.siglip_model = from_pretrained(automodel_name, siglip_checkpoint, lora=adapters)
        self.hidden_size = dim
        assert self.hidden_size == self.siglip_model.vcfg["hidden_size"]
        assert self.hidden_size == self.siglip_model.tcfg["hidden_size"]
        self.num_patches = (self.image_size // self.patch_size) ** 2
        if self.siglip_model.grid ** 2 != self.num_patches:
            raise ValueError("image_size / patch_size do not match the SigLIP vision tower")
        if not lora:
            self.frozen_submodule("siglip_model")
        self.text_token = nn.Parameter(torch.randn(1, 1, dim))
        self.image_token = nn.Parameter(torch.randn(1, 1, dim))
        self.pick_place = Components.get_by_name(pick_place_model, dim=dim, patch_size=patch_size,
                                                 num_patches=self.num_patches, **kwargs)
        self.pick_place.fusion.compute_dtype = self.compute_dtype
        self._train_arena: Optional[ParamArena] = None
        self._grad_hooks: List[Callable[[int], None]] = []

    # ---- same runtime plumbing as RGBOnly ---------------------------------------------------
    def trainable_params(self):
        return [(n, p) for n, p in self.named_parameters() if "siglip_model" not in n or ".lora_" in n]

    def trainable_arena(self) -> ParamArena:
        a = self._train_arena
        if a is None or not a.still_bound():
            named = self.trainable_params()
            a = arena_of([p for _, p in named])
            if a is None:
                a = ParamArena(named, with_grad=True, with_shadow=True)
            self._train_arena = a
        return a

    def register_grad_ready_hook(self, fn: Callable[[int], None]) -> None:
        self._grad_hooks.append(fn)

    def _grad_ready(self, layer: int) -> None:
        for fn in self._grad_hooks:
            fn(layer)

    # ---- forward ------------------------------------------------------------------------------
    def _towers(self, x):
        rgb = x["rgb"]
        if not rgb.is_cuda:
            raise RuntimeError(f"{type(self).__name__} runs on the MI355X only (there is no CPU path; the CPU "
                               f"restatement lives in oracle/ for tests)")
        cdt = self.compute_dtype
        self.trainable_arena().sync_shadow()
        self.siglip_model.begin_step()
        self.pick_place.collapse_async()                  # decoder collapse beside the towers (side stream)
        img = self._vision(x, cdt)                                            # [B*n, W] fp32
        txt = self.siglip_model.text_tokens(x["instruction"], cdt)            # [B*L, W] fp32
        return img, txt

    def _vision(self, x, cdt):
        return self.siglip_model.vision_tokens(x["rgb"], cdt)

    def _segments(self, x, img, txt):
        b = x["rgb"].shape[0]
        n_txt = txt.shape[0] // b
        n_img = self.num_patches
        d = self.hidden_size
        return [dict(src=txt, bstride=n_txt * d, n=n_txt, batch=b, lead=self.text_token, pos=None, modality=0,
                     role="txt"),
                dict(src=img, bstride=n_img * d, n=n_img, batch=b, lead=self.image_token, pos=None, modality=1,
                     role="img")], None

    # ---- LoRA: the tail's backward continues into the towers -----------------------------------
    def tower_backward(self, dx: torch.Tensor, segments, batch: int, seq: int) -> None:
        """dx fp32 [B*seq, dim] = d loss / d (assembled fusion input).  A segment's tower tokens sit
        one row after its lead token and enter the sum unscaled, so their gradient is that slice."""
        if not self.lora:
            return
        d = self.hidden_size
        dxv = dx.view(batch, seq, d)
        by_role = {sg["role"]: dxv[:, sg["row_off"] + 1: sg["row_off"] + 1 + sg["n"]] for sg in segments}
        g = self.trainable_arena().grad_view
        self.siglip_model.backward("vision_model", self._vision_grad(by_role, batch), g)
        self.siglip_model.backward("text_model", by_role["txt"].reshape(-1, d), g)

    def _vision_grad(self, by_role, batch: int) -> torch.Tensor:
        return by_role["img"].reshape(-1, self.hidden_size)

    def forward(self, x):
        img, txt = self._towers(x)
        segments, row_mask = self._segments(x, img, txt)
        params = [p for _, p in self.trainable_params()]
        heats = SegmentTailFn.apply(self, segments, row_mask, *params)
        out = {"attn_weights": None}
        for name, h in zip(self.pick_place.heads_out, heats):
            out[f"{name}_heatmap"] = h
        if self.pick_place.has_mask:                      # models/pickplace.py:66-69
            out["mask_heatmap"] = heats[len(self.pick_place.heads_out)]
        return out


class SiglipSequential(SigLip):
    def __init__(self, context_length, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.context_length = context_length
        self.context_pos_embedding = nn.Parameter(
            torch.randn(1, context_length * (self.num_patches + 1), self.hidden_size))

    def _segments(self, x, img, txt):
        ctx = x["rgb_context"]
        b, t = ctx.shape[:2]
        if t != self.context_length:
            raise ValueError(f"rgb_context has {t} frames, model was built with context_length={self.context_length}")
        n, d = self.num_patches, self.hidden_size
        if self.lora:
            # ONE pass of the vision tower over [current image | context frames]: the reference runs the
            # (LoRA-wrapped) vision_model twice (models/siglip.py:57,102-104); images are independent, so
            # the tokens are the same and the adapters' gradient is accumulated by a single backward
            cf = img[b * n:].view(b, t, n, d)
            img = img[:b * n]
        else:
            cf = self.siglip_model.vision_tokens(ctx.flatten(0, 1), self.compute_dtype).view(b, t, n, d)
        (seg_txt, seg_img), _ = super()._segments(x, img, txt)
        segs = [seg_txt]
        for f in range(t):        # frame f: src offset f*n*d inside each sample's T*n*d block
            segs.append(dict(src=cf[:, f], bstride=t * n * d, n=n, batch=b, lead=self.image_token,
                             pos=(self.context_pos_embedding, f * (n + 1)), modality=1, role=("ctx", f)))
        segs.append(seg_img)
        # query-row mask [ones | ctx_mask repeated n+1 | ones]  (models/siglip.py:118-130)
        m = x["context_attention_mask"].to(torch.int32)
        row_mask = torch.cat([torch.ones(b, seg_txt["n"] + 1, dtype=torch.int32, device=m.device),
                              m.repeat_interleave(n + 1, dim=1),
                              torch.ones(b, n + 1, dtype=torch.int32, device=m.device)], dim=1).contiguous()
        return segs, row_mask

    def _vision(self, x, cdt):
        if not self.lora:
            return super()._vision(x, cdt)
        return self.siglip_model.vision_tokens(torch.cat([x["rgb"], x["rgb_context"].flatten(0, 1)], dim=0), cdt)

    def _vision_grad(self, by_role, batch: int) -> torch.Tensor:
        d, t = self.hidden_size, self.context_length
        ctx = torch.stack([by_role[("ctx", f)] for f in range(t)], dim=1)       # [B, T, n, d]
        return torch.cat([by_role["img"].reshape(-1, d), ctx.reshape(-1, d)], dim=0)

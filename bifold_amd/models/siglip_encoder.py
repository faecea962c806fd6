"""Frozen SigLIP image/text towers on libbifold_hip.

The reference obtains them from `transformers.AutoModel.from_pretrained(automodel_name)`
(bifold/models/siglip.py:5,26) and uses only `vision_model(...).last_hidden_state` and
`text_model(...).last_hidden_state` (:59-60,65,102-104): the pooled heads and logits that HF
also computes are discarded, so they are not computed here.  Parameter names are HF's
(`vision_model.embeddings.patch_embedding.weight`, `text_model.encoder.layers.{i}.self_attn.q_proj.weight`
...; full list pinned by tests/golden/tiny_siglip.npz), so a SiglipModel state dict loads with
`load_state_dict`.  Block = pre-LN (eps 1e-6), q/k/v/out projections with bias, tanh-GELU MLP; no
class token, no attention mask.  Forward-only (the towers are frozen when lora=false).
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import torch
from torch import nn

from ..hip import lib as L
from ..hip import ops
from .arena import ParamArena
from .params import add_param, layernorm_init, linear_init

_BASE = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12)
# Hyper-parameters of the published checkpoints the reference's configs name
# (conf/model/siglip*.yaml: google/siglip-base-patch16-{image_size}).
SIGLIP_CONFIGS: Dict[str, Dict] = {
    f"google/siglip-base-patch16-{s}": dict(vision=dict(_BASE, image_size=s, patch_size=16),
                                            text=dict(_BASE, vocab_size=32000, max_position_embeddings=64))
    for s in (224, 256, 384, 512)
}
LN_EPS = 1e-6


class SiglipTowers(nn.Module):
    def __init__(self, vision: Dict, text: Dict):
        super().__init__()
        self.vcfg, self.tcfg = dict(vision), dict(text)
        vw, tw = vision["hidden_size"], text["hidden_size"]
        p = vision["patch_size"]
        self.grid = vision["image_size"] // p
        self.patch = p
        n = self.grid ** 2
        for k, v in linear_init(vw, 3 * p * p).items():
            add_param(self, f"vision_model.embeddings.patch_embedding.{k}", v.view(vw, 3, p, p) if k == "weight" else v)
        add_param(self, "vision_model.embeddings.position_embedding.weight", torch.randn(n, vw) * vw ** -0.5)
        self._layers("vision_model", vision)
        for k, v in layernorm_init(vw).items():
            add_param(self, f"vision_model.post_layernorm.{k}", v)
        # attention-pooling head: part of the checkpoint schema, never evaluated (its output is unused)
        add_param(self, "vision_model.head.probe", torch.randn(1, 1, vw))
        add_param(self, "vision_model.head.attention.in_proj_weight", torch.randn(3 * vw, vw) * vw ** -0.5)
        add_param(self, "vision_model.head.attention.in_proj_bias", torch.zeros(3 * vw))
        for k, v in linear_init(vw, vw).items():
            add_param(self, f"vision_model.head.attention.out_proj.{k}", v)
        for k, v in layernorm_init(vw).items():
            add_param(self, f"vision_model.head.layernorm.{k}", v)
        for k, v in linear_init(vision["intermediate_size"], vw).items():
            add_param(self, f"vision_model.head.mlp.fc1.{k}", v)
        for k, v in linear_init(vw, vision["intermediate_size"]).items():
            add_param(self, f"vision_model.head.mlp.fc2.{k}", v)
        add_param(self, "text_model.embeddings.token_embedding.weight", torch.randn(text["vocab_size"], tw) * tw ** -0.5)
        add_param(self, "text_model.embeddings.position_embedding.weight",
                  torch.randn(text["max_position_embeddings"], tw) * tw ** -0.5)
        self._layers("text_model", text)
        for k, v in layernorm_init(tw).items():
            add_param(self, f"text_model.final_layer_norm.{k}", v)
        for k, v in linear_init(tw, tw).items():
            add_param(self, f"text_model.head.{k}", v)
        add_param(self, "logit_scale", torch.zeros(1))
        add_param(self, "logit_bias", torch.zeros(1))
        self._arena: Optional[ParamArena] = None
        self._packed: Dict[str, torch.Tensor] = {}
        self._packed_version = -1
        self._err_flag: Optional[torch.Tensor] = None

    def _layers(self, tower: str, cfg: Dict):
        w, inter = cfg["hidden_size"], cfg["intermediate_size"]
        for i in range(cfg["num_hidden_layers"]):
            b = f"{tower}.encoder.layers.{i}."
            for ln in ("layer_norm1", "layer_norm2"):
                for k, v in layernorm_init(w).items():
                    add_param(self, f"{b}{ln}.{k}", v)
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):          # HF's registration order
                for k, v in linear_init(w, w).items():
                    add_param(self, f"{b}self_attn.{proj}.{k}", v)
            for k, v in linear_init(inter, w).items():
                add_param(self, f"{b}mlp.fc1.{k}", v)
            for k, v in linear_init(w, inter).items():
                add_param(self, f"{b}mlp.fc2.{k}", v)

    # ------------------------------------------------------------------ runtime
    def _runtime(self, cdt: torch.dtype) -> ParamArena:
        a = self._arena
        if a is None or not a.still_bound() or (cdt != torch.float32 and a.shadow is None):
            a = ParamArena(self.named_parameters(), with_grad=False, with_shadow=(cdt != torch.float32))
            self._arena = a
            self._packed_version = -1
        a.sync_shadow()
        ver = sum(p._version for p in a.params)
        if ver != self._packed_version or self._packed.get("dtype") != cdt:
            # q/k/v live as three parameters (checkpoint schema) but run as ONE in-projection GEMM
            # that writes the packed [B, L, 3, H, D] buffer the attention kernels read
            self._packed = {"dtype": cdt}
            for tower, cfg in (("vision_model", self.vcfg), ("text_model", self.tcfg)):
                for i in range(cfg["num_hidden_layers"]):
                    b = f"{tower}.encoder.layers.{i}.self_attn."
                    ws = [a.operand(self.get_parameter(f"{b}{n}_proj.weight"), cdt) for n in "qkv"]
                    bs = [self.get_parameter(f"{b}{n}_proj.bias").data for n in "qkv"]
                    self._packed[b + "w"] = torch.cat(ws, dim=0).contiguous()
                    self._packed[b + "b"] = torch.cat(bs, dim=0).contiguous()
            self._packed_version = ver
        return a

    def _w(self, name: str, cdt):
        return self._arena.operand(self.get_parameter(name), cdt)

    def _f(self, name: str):
        return self.get_parameter(name).data

    def _encoder(self, x, tower: str, cfg: Dict, batch: int, seq: int, cdt):
        heads = cfg["num_attention_heads"]
        d = cfg["hidden_size"] // heads
        for i in range(cfg["num_hidden_layers"]):
            b = f"{tower}.encoder.layers.{i}."
            h, _, _ = ops.layernorm_fwd(x, self._f(b + "layer_norm1.weight"), self._f(b + "layer_norm1.bias"), cdt,
                                        eps=LN_EPS)
            qkv = ops.gemm(h, self._packed[b + "self_attn.w"], bias=self._packed[b + "self_attn.b"])
            o, _ = ops.attention_fwd(qkv, batch, seq, heads, d, d ** -0.5, L.MASK_NONE)
            ops.gemm(o, self._w(b + "self_attn.out_proj.weight", cdt), bias=self._f(b + "self_attn.out_proj.bias"),
                     residual=x, out=x)
            h, _, _ = ops.layernorm_fwd(x, self._f(b + "layer_norm2.weight"), self._f(b + "layer_norm2.bias"), cdt,
                                        eps=LN_EPS)
            a = ops.gemm(h, self._w(b + "mlp.fc1.weight", cdt), bias=self._f(b + "mlp.fc1.bias"),
                         act=L.ACT_TANH_GELU)
            ops.gemm(a, self._w(b + "mlp.fc2.weight", cdt), bias=self._f(b + "mlp.fc2.bias"), residual=x, out=x)
        return x

    @torch.no_grad()
    def vision_tokens(self, img: torch.Tensor, cdt: torch.dtype) -> torch.Tensor:
        """[N,3,S,S] fp32 -> post_layernorm(tokens) fp32 [N*(S/p)^2, W]."""
        self._runtime(cdt)
        nimg, _, size, _ = img.shape
        if size != self.grid * self.patch:
            raise ValueError(f"image size {size} != {self.grid * self.patch} expected by the SigLIP vision tower")
        n = self.grid * self.grid
        vw = self.vcfg["hidden_size"]
        cols = ops.patchify(img.contiguous(), self.patch, cdt)
        wconv = self._w("vision_model.embeddings.patch_embedding.weight", cdt).view(vw, -1)
        x = torch.empty(nimg * n, vw, dtype=torch.float32, device=img.device)
        # conv bias and the learned position table ride in the GEMM epilogue (table row = patch index)
        ops.gemm(cols, wconv, bias=self._f("vision_model.embeddings.patch_embedding.bias"),
                 residual=self._f("vision_model.embeddings.position_embedding.weight"), resid_mod=n, out=x)
        x = self._encoder(x, "vision_model", self.vcfg, nimg, n, cdt)
        y, _, _ = ops.layernorm_fwd(x, self._f("vision_model.post_layernorm.weight"),
                                    self._f("vision_model.post_layernorm.bias"), torch.float32, eps=LN_EPS)
        return y

    @torch.no_grad()
    def text_tokens(self, ids: torch.Tensor, cdt: torch.dtype) -> torch.Tensor:
        """[B,L] int -> final_layer_norm(tokens) fp32 [B*L, W]."""
        self._runtime(cdt)
        ids = ids.to(torch.int32).contiguous()
        batch, ctx = ids.shape
        if ctx > self.tcfg["max_position_embeddings"]:
            raise ValueError(f"instruction length {ctx} exceeds the SigLIP text tower's "
                             f"{self.tcfg['max_position_embeddings']} positions")
        if self._err_flag is None or self._err_flag.device != ids.device:
            self._err_flag = torch.zeros(1, dtype=torch.int32, device=ids.device)
        x = ops.embed_tokens(ids, self._f("text_model.embeddings.token_embedding.weight"),
                             self._f("text_model.embeddings.position_embedding.weight"), self._err_flag)
        x = self._encoder(x, "text_model", self.tcfg, batch, ctx, cdt)
        y, _, _ = ops.layernorm_fwd(x, self._f("text_model.final_layer_norm.weight"),
                                    self._f("text_model.final_layer_norm.bias"), torch.float32, eps=LN_EPS)
        return y

    def check_token_ids(self) -> None:
        if self._err_flag is not None and int(self._err_flag.item()) != 0:
            self._err_flag.zero_()
            raise IndexError("instruction contains a token id outside the SigLIP vocabulary")


def from_pretrained(name: str, checkpoint: Optional[str] = None) -> SiglipTowers:
    """Stand-in for AutoModel.from_pretrained (no network): build the towers from the named
    checkpoint's hyper-parameters and fill them from a local copy when one is given: the
    Hugging Face snapshot directory, its model.safetensors, or a state dict with the same
    (transformers SiglipModel) key names -- see checkpoint_io.read_state_dict."""
    if name not in SIGLIP_CONFIGS:
        raise ValueError(f"unknown SigLIP model {name!r}; known: {sorted(SIGLIP_CONFIGS)}")
    cfg = SIGLIP_CONFIGS[name]
    model = SiglipTowers(cfg["vision"], cfg["text"])
    if checkpoint:
        from .checkpoint_io import load_into, read_state_dict
        load_into(model, read_state_dict(checkpoint), f"SigLIP {name}")
    else:
        warnings.warn(f"SigLIP {name}: no checkpoint given, using random initialisation "
                      f"(set model.siglip_checkpoint=<snapshot dir | model.safetensors | state_dict.pt>)")
    return model.eval()

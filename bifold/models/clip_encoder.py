"""Frozen CLIP image/text towers on libbifold_hip.

Parameter names follow OpenAI CLIP's state dict (the reference vendors it under
bifold/models/clip/model.py; key list in SURVEY.md §8b), so a CLIP checkpoint loads with
`load_state_dict`.  Only what BiFold uses is computed: per-token embeddings of both towers
(`forward_with_embeddings` model.py:296-317, `encode_text_with_embeddings` :447-461) --
no pooling, no output projections, no logits.  Both towers are frozen in BiFold
(models/rgb_clip.py:31), so this file is forward-only and keeps no activations.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, Optional

import torch
from torch import nn

from bifold_amd.hip import lib as L
from bifold_amd.hip import ops
from .arena import ParamArena
from .params import add_param, layernorm_init

# Hyper-parameters that `build_model` (clip/model.py:509-556) infers from the published
# checkpoints of each name in clip/__init__.py:_MODELS (ViT variants only: the ResNet
# towers are reachable only through the out-of-scope text_unet model).
CLIP_CONFIGS: Dict[str, Dict[str, int]] = {
    "ViT-B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                     vision_patch_size=32, context_length=77, vocab_size=49408, transformer_width=512,
                     transformer_heads=8, transformer_layers=12),
    "ViT-B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                     vision_patch_size=16, context_length=77, vocab_size=49408, transformer_width=512,
                     transformer_heads=8, transformer_layers=12),
    "ViT-L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024,
                     vision_patch_size=14, context_length=77, vocab_size=49408, transformer_width=768,
                     transformer_heads=12, transformer_layers=12),
    "ViT-L/14@336px": dict(embed_dim=768, image_resolution=336, vision_layers=24, vision_width=1024,
                           vision_patch_size=14, context_length=77, vocab_size=49408,
                           transformer_width=768, transformer_heads=12, transformer_layers=12),
}
_MODELS = CLIP_CONFIGS  # the name the reference asserts against (rgb_clip.py:29)


class CLIPEncoder(nn.Module):
    def __init__(self, *, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        self.cfg = dict(embed_dim=embed_dim, image_resolution=image_resolution, vision_layers=vision_layers,
                        vision_width=vision_width, vision_patch_size=vision_patch_size,
                        context_length=context_length, vocab_size=vocab_size,
                        transformer_width=transformer_width, transformer_heads=transformer_heads,
                        transformer_layers=transformer_layers)
        vw, tw, p = vision_width, transformer_width, vision_patch_size
        g = image_resolution // p
        self.grid = g
        self.vision_heads = vw // 64                      # clip/model.py:351
        self.text_heads = transformer_heads
        self.vision_layers, self.text_layers = vision_layers, transformer_layers
        self.vision_width, self.text_width = vw, tw
        self.patch = p
        self.context_length = context_length

        def rn(*shape, std):
            return torch.randn(*shape) * std

        # same initial scales as CLIP.__init__/initialize_parameters (clip/model.py:250-262,380-412)
        fan = 3 * p * p
        add_param(self, "visual.conv1.weight", torch.empty(vw, 3, p, p).uniform_(-fan ** -0.5, fan ** -0.5))
        add_param(self, "visual.class_embedding", rn(vw, std=vw ** -0.5))
        add_param(self, "visual.positional_embedding", rn(g * g + 1, vw, std=vw ** -0.5))
        add_param(self, "visual.proj", rn(vw, embed_dim, std=vw ** -0.5))
        for ln in ("ln_pre", "ln_post"):
            for k, v in layernorm_init(vw).items():
                add_param(self, f"visual.{ln}.{k}", v)
        self._blocks("visual.transformer.resblocks", vw, vision_layers)
        self._blocks("transformer.resblocks", tw, transformer_layers)
        add_param(self, "token_embedding.weight", rn(vocab_size, tw, std=0.02))
        add_param(self, "positional_embedding", rn(context_length, tw, std=0.01))
        for k, v in layernorm_init(tw).items():
            add_param(self, f"ln_final.{k}", v)
        add_param(self, "text_projection", rn(tw, embed_dim, std=tw ** -0.5))
        add_param(self, "logit_scale", torch.tensor(math.log(1 / 0.07)))
        self.transformer.width = tw                          # read by RGBOnly (rgb_clip.py:32)
        self.visual.output_width = vw
        self._arena: Optional[ParamArena] = None
        self._err_flag: Optional[torch.Tensor] = None
        # dtype: fp8 (BASELINE configs[4]): the LayerNorm -> in-projection and LayerNorm -> c_fc pairs of both towers
        # run in OCP e4m3 with per-token / per-channel scales (csrc/gemm_fp8.hip); everything else stays bf16
        self.fp8 = False
        self._w8_cache: Dict[str, tuple] = {}

    def _w8(self, name: str):
        """(e4m3 bytes [N, K], fp32 per-channel scales [N]) of a frozen weight, quantised once per load."""
        p = self.get_parameter(name)
        hit = self._w8_cache.get(name)
        if hit is None or hit[2] != p._version or hit[0].device != p.device:
            q, sc = ops.quantize_rows_fp8(p.data.view(p.shape[0], -1))
            hit = (q, sc, p._version)
            self._w8_cache[name] = hit
        return hit[0], hit[1]

    def _blocks(self, prefix: str, width: int, layers: int):
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        fc_std = (2 * width) ** -0.5
        bound = width ** -0.5
        for i in range(layers):
            b = f"{prefix}.{i}."
            add_param(self, b + "attn.in_proj_weight", torch.randn(3 * width, width) * width ** -0.5)
            add_param(self, b + "attn.in_proj_bias", torch.zeros(3 * width))
            add_param(self, b + "attn.out_proj.weight", torch.randn(width, width) * proj_std)
            add_param(self, b + "attn.out_proj.bias", torch.zeros(width))
            for ln in ("ln_1", "ln_2"):
                for k, v in layernorm_init(width).items():
                    add_param(self, f"{b}{ln}.{k}", v)
            add_param(self, b + "mlp.c_fc.weight", torch.randn(4 * width, width) * fc_std)
            add_param(self, b + "mlp.c_fc.bias", torch.empty(4 * width).uniform_(-bound, bound))
            add_param(self, b + "mlp.c_proj.weight", torch.randn(width, 4 * width) * proj_std)
            add_param(self, b + "mlp.c_proj.bias", torch.empty(width).uniform_(-0.5 * bound, 0.5 * bound))

    # ------------------------------------------------------------------ runtime
    def _runtime(self, cdt: torch.dtype) -> ParamArena:
        a = self._arena
        if a is None or not a.still_bound() or (cdt != torch.float32 and a.shadow is None):
            a = ParamArena(self.named_parameters(), with_grad=False, with_shadow=(cdt != torch.float32),
                           shadow_dtype=cdt if cdt != torch.float32 else torch.bfloat16)
            self._arena = a
        if cdt != torch.float32:
            a.set_shadow_dtype(cdt)
        a.sync_shadow()
        return a

    def _w(self, name: str, cdt: torch.dtype) -> torch.Tensor:
        return self._arena.operand(self.get_parameter(name), cdt)

    def _f(self, name: str) -> torch.Tensor:
        return self.get_parameter(name).data

    def _tower(self, x: torch.Tensor, prefix: str, layers: int, heads: int, batch: int, seq: int,
               cdt: torch.dtype, causal: bool, final=None, h0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ResidualAttentionBlock x layers (clip/model.py:192-219) on the fp32 residual stream
        x [B*L, W]; every GEMM is fused with its bias / QuickGELU epilogue.  `final` = (name of the LayerNorm
        that follows the last block, its output dtype): when given, the tower returns that LayerNorm's output."""
        width = x.shape[1]
        d = width // heads
        mode = L.MASK_CAUSAL if causal else L.MASK_NONE
        fp8 = self.fp8 and cdt == torch.bfloat16 and width % 128 == 0      # the e4m3 MFMA consumes K in steps of 128
        # bf16 mode: the two residual adds of a block move OUT of the GEMM epilogues into the LayerNorm that follows
        # them (bf_add_layernorm_fwd): out-proj and c_proj then run the plain-bf16 epilogue -- no fp32 residual tile
        # to fetch before the first MFMA, half the bytes to store -- and the LayerNorm, which streams the residual
        # anyway, adds the branch's bf16 output on the way (in-step: 139 + 43 -> ~170 us for out-proj + ln_2 at the
        # vision tower's shape).  `final` = the LayerNorm after the last block (ln_post / ln_final), fused the same way.
        fuse_add = cdt in ops.HALF and not fp8 and width in (256, 512, 768, 1024) and final is not None
        lazy = fuse_add                   # the frozen tower writes its stream ONCE per block (round 5: +0.8 % on the step, same bits)
        h = None if fp8 else h0           # ln_1 of block 0, when the caller has already computed it (K1)
        for i in range(layers):
            b = f"{prefix}.{i}."
            if fp8:
                h8, hs = ops.layernorm_fwd_fp8(x, self._f(b + "ln_1.weight"), self._f(b + "ln_1.bias"))
                w8, ws = self._w8(b + "attn.in_proj_weight")
                qkv = ops.gemm_fp8(h8, hs, w8, ws, bias=self._f(b + "attn.in_proj_bias"))
            else:
                if h is None:
                    h, _, _ = ops.layernorm_fwd(x, self._f(b + "ln_1.weight"), self._f(b + "ln_1.bias"), cdt)
                qkv = ops.gemm(h, self._w(b + "attn.in_proj_weight", cdt), bias=self._f(b + "attn.in_proj_bias"))
                h = None
            o, _ = ops.attention_fwd(qkv, batch, seq, heads, d, d ** -0.5, mode)
            if fuse_add:
                # The towers are frozen: nothing reads the stream between a block's two branches, so it is written ONCE
                # per block (round 5).  The pass after the attention branch only produces ln_2(x + attn) (no x_out: 8
                # bytes per element instead of 12); the pass after the MLP adds BOTH branches to the block's input,
                # (x + attn) + mlp in that order -- the bits of the two-pass form -- and writes the stream (14 instead
                # of 12).  (The two read-modify-write passes it replaced: removed in round 6.)
                attn = ops.gemm(o, self._w(b + "attn.out_proj.weight", cdt), bias=self._f(b + "attn.out_proj.bias"))
                h, _, _, _ = ops.add_layernorm_fwd(x, attn, self._f(b + "ln_2.weight"), self._f(b + "ln_2.bias"), cdt,
                                                   write_x=not lazy)
                a = ops.gemm(h, self._w(b + "mlp.c_fc.weight", cdt), bias=self._f(b + "mlp.c_fc.bias"),
                             act=L.ACT_QUICK_GELU)
                delta = ops.gemm(a, self._w(b + "mlp.c_proj.weight", cdt), bias=self._f(b + "mlp.c_proj.bias"))
                first, second = (attn, delta) if lazy else (delta, None)
                if i + 1 < layers:
                    nb = f"{prefix}.{i + 1}."
                    h, _, _, _ = ops.add_layernorm_fwd(x, first, self._f(nb + "ln_1.weight"), self._f(nb + "ln_1.bias"), cdt,
                                                       delta2=second)
                else:
                    # the last block: nothing reads the stream after the final LayerNorm
                    gname, out_dtype = final
                    y, _, _, _ = ops.add_layernorm_fwd(x, first, self._f(gname + ".weight"), self._f(gname + ".bias"),
                                                       out_dtype, delta2=second, write_x=not lazy)
                    return y
                continue
            ops.gemm(o, self._w(b + "attn.out_proj.weight", cdt), bias=self._f(b + "attn.out_proj.bias"),
                     residual=x, out=x)
            if fp8:
                h8, hs = ops.layernorm_fwd_fp8(x, self._f(b + "ln_2.weight"), self._f(b + "ln_2.bias"))
                w8, ws = self._w8(b + "mlp.c_fc.weight")
                a = ops.gemm_fp8(h8, hs, w8, ws, bias=self._f(b + "mlp.c_fc.bias"), act=L.ACT_QUICK_GELU)
            else:
                h2, _, _ = ops.layernorm_fwd(x, self._f(b + "ln_2.weight"), self._f(b + "ln_2.bias"), cdt)
                a = ops.gemm(h2, self._w(b + "mlp.c_fc.weight", cdt), bias=self._f(b + "mlp.c_fc.bias"),
                             act=L.ACT_QUICK_GELU)
            ops.gemm(a, self._w(b + "mlp.c_proj.weight", cdt), bias=self._f(b + "mlp.c_proj.bias"),
                     residual=x, out=x)
        if final is not None:
            gname, out_dtype = final
            y, _, _ = ops.layernorm_fwd(x, self._f(gname + ".weight"), self._f(gname + ".bias"), out_dtype)
            return y
        return x

    @torch.no_grad()
    def encode_image_with_embeddings(self, img: torch.Tensor, cdt: torch.dtype, out_dtype=None):
        """[B,3,S,S] fp32 -> ln_post(tokens) [B*(g*g+1), W] (clip/model.py:296-317)."""
        self._runtime(cdt)
        batch, _, size, _ = img.shape
        if size != self.grid * self.patch:
            raise ValueError(f"image size {size} != {self.grid * self.patch} expected by the vision tower")
        n = self.grid * self.grid
        vw = self.vision_width
        cols = ops.patchify(img.contiguous(), self.patch, cdt)
        wconv = self._w("visual.conv1.weight", cdt).view(vw, -1)
        if cols.shape[1] != wconv.shape[1]:               # patch 14: K padded 588 -> 640, zero columns
            key = (cdt, cols.shape[1], self.get_parameter("visual.conv1.weight")._version, wconv.data_ptr())
            if getattr(self, "_wconv_pad_key", None) != key:
                pad = torch.zeros(vw, cols.shape[1], dtype=cdt, device=wconv.device)
                pad[:, :wconv.shape[1]] = wconv
                self._wconv_pad, self._wconv_pad_key = pad, key
            wconv = self._wconv_pad
        pos = self._f("visual.positional_embedding")
        fp8 = self.fp8 and cdt == torch.bfloat16 and vw % 128 == 0
        h0 = None
        if not fp8:
            # K1: the patch-embed GEMM's epilogue adds pos[1 + p] and writes row b*n + p of the
            # [B*n] problem to row b*(n+1) + 1 + p of the token stream; one pass then sets the class rows, applies
            # ln_pre in place and emits block 0's ln_1 -- no [B*n, W] patch buffer, no separate assembly / ln_1 pass
            x = torch.empty(batch * (n + 1), vw, dtype=torch.float32, device=img.device)
            ops.gemm(cols, wconv, residual=pos[1:], resid_mod=n, out=x, rows_in=n, rows_out=n + 1, row_off=1)
            b0 = "visual.transformer.resblocks.0."
            h0 = ops.vision_lnpre_ln1(x, self._f("visual.class_embedding"), pos[0], self._f("visual.ln_pre.weight"),
                                      self._f("visual.ln_pre.bias"), self._f(b0 + "ln_1.weight"),
                                      self._f(b0 + "ln_1.bias"), cdt, batch, n)
        else:
            patches = ops.gemm(cols, wconv, out_dtype=torch.float32)
            x = ops.vision_assemble(patches, self._f("visual.class_embedding"), pos, self._f("visual.ln_pre.weight"),
                                    self._f("visual.ln_pre.bias"), batch, n)
        return self._tower(x, "visual.transformer.resblocks", self.vision_layers, self.vision_heads, batch,
                           n + 1, cdt, causal=False, final=("visual.ln_post", out_dtype or cdt), h0=h0)

    @torch.no_grad()
    def encode_text_with_embeddings(self, tokens: torch.Tensor, cdt: torch.dtype):
        """[B,ctx] int -> ln_final(tokens) fp32 [B*ctx, W] (clip/model.py:447-461)."""
        self._runtime(cdt)
        if tokens.dtype != torch.int32:
            tokens = tokens.to(torch.int32)
        tokens = tokens.contiguous()
        batch, ctx = tokens.shape
        if ctx != self.context_length:
            raise ValueError(f"instruction length {ctx} != CLIP context length {self.context_length}")
        if self._err_flag is None or self._err_flag.device != tokens.device:
            self._err_flag = torch.zeros(1, dtype=torch.int32, device=tokens.device)
        x = ops.embed_tokens(tokens, self._f("token_embedding.weight"), self._f("positional_embedding"),
                             self._err_flag)
        return self._tower(x, "transformer.resblocks", self.text_layers, self.text_heads, batch, ctx, cdt,
                           causal=True, final=("ln_final", torch.float32))

    def check_token_ids(self) -> None:
        """Raise if any forward since the last call saw a token id outside the vocabulary
        (device flag; reading it synchronises, so it is not called on the hot path)."""
        if self._err_flag is not None and int(self._err_flag.item()) != 0:
            self._err_flag.zero_()
            raise IndexError("instruction contains a token id outside the CLIP vocabulary")


def load(name: str, device="cpu", checkpoint: Optional[str] = None, **_):
    """Stand-in for clip.load (clip/__init__.py:117-176), which downloads weights: there is
    no network here, so the tower is built from its hyper-parameters and filled from
    `checkpoint` when one is given: OpenAI's TorchScript archive for that name, or a plain
    state dict (checkpoint_io.read_state_dict lists the accepted files)."""
    if name not in CLIP_CONFIGS:
        raise RuntimeError(f"Model {name} not found; available models = {list(CLIP_CONFIGS)}")
    model = CLIPEncoder(**CLIP_CONFIGS[name])
    if checkpoint:
        from .checkpoint_io import load_into, read_state_dict
        load_into(model, read_state_dict(checkpoint), f"CLIP {name}")
    else:
        warnings.warn(f"CLIP {name}: no checkpoint given, using random initialisation "
                      f"(set model.clip_checkpoint=<ViT-*.pt | state_dict.pt> to load weights)")
    return model.eval(), None

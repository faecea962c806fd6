"""Functional fp32 CPU restatement of BiFold's VLM training-step hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the product package
never imports this file.

Style: every function is a pure function of a flat ``state`` mapping
(reference state-dict key -> tensor) plus inputs.  There is no nn.Module tree;
the reference's state-dict key names ARE the interface, which is what lets the
same checkpoint feed the reference, this oracle and the HIP path.

Each function cites the reference lines it restates (paths relative to
/root/reference/bifold/).  Layout here is batch-first ``[B, L, W]`` throughout;
the reference runs its CLIP towers sequence-first, which only permutes memory.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

State = Mapping[str, torch.Tensor]


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """nn.LayerNorm over the last dim, biased variance, fp32
    (models/clip/model.py:178-184; models/modules.py:11)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def quick_gelu(x: torch.Tensor):
    """x * sigmoid(1.702 x) (models/clip/model.py:187-189)."""
    return x * torch.sigmoid(1.702 * x)


def erf_gelu(x: torch.Tensor):
    """nn.GELU() default = exact erf form (models/modules.py:23)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def split_heads(t: torch.Tensor, heads: int):
    b, l, w = t.shape
    return t.reshape(b, l, heads, w // heads).permute(0, 2, 1, 3)  # [B,h,L,d]


def merge_heads(t: torch.Tensor):
    b, h, l, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, l, h * d)


def _drop(x: torch.Tensor, spec):
    """nn.Dropout with a GIVEN mask: spec = (p, keep mask broadcastable to x) or None.  The reference draws its masks from
    torch's generator; a checker feeds the masks the device generated (bf_dropout_mask) instead."""
    if spec is None:
        return x
    p, keep = spec
    return x * keep.to(x.dtype).reshape(x.shape) / (1.0 - p)


def attention_core(q, k, v, scale: float, causal: bool = False,
                   query_row_mask: Optional[torch.Tensor] = None, p_drop=None):
    """softmax(q k^T * scale [+mask]) v on [B,h,L,d] tensors.

    * causal: additive -inf above the diagonal (models/clip/model.py:414-420).
    * query_row_mask [B,L] (0 = masked): the reference fills the WHOLE score row
      of a masked *query* with -1e5, not the key column
      (models/transformer.py:33-35) -> that row's softmax becomes uniform.
    """
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if causal:
        l = s.shape[-1]
        upper = torch.ones(l, l, dtype=torch.bool, device=s.device).triu(1)
        s = s.masked_fill(upper, float("-inf"))
    if query_row_mask is not None:
        s = s.masked_fill(query_row_mask[:, None, :, None] == 0, -100000.0)
    p = _drop(torch.softmax(s, dim=-1), p_drop)              # models/transformer.py:36: dropout AFTER the softmax
    return torch.matmul(p, v)


# --------------------------------------------------------------------------
# CLIP towers (frozen)
# --------------------------------------------------------------------------
def clip_resblock(state: State, pfx: str, x: torch.Tensor, heads: int, causal: bool):
    """ResidualAttentionBlock (models/clip/model.py:192-219):
    x += MHA(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))."""
    w = x.shape[-1]
    h = layer_norm(x, state[pfx + "ln_1.weight"], state[pfx + "ln_1.bias"])
    qkv = F.linear(h, state[pfx + "attn.in_proj_weight"], state[pfx + "attn.in_proj_bias"])
    q, k, v = (split_heads(t, heads) for t in qkv.split(w, dim=-1))
    o = attention_core(q, k, v, scale=(w // heads) ** -0.5, causal=causal)
    o = F.linear(merge_heads(o), state[pfx + "attn.out_proj.weight"],
                 state[pfx + "attn.out_proj.bias"])
    x = x + o
    h = layer_norm(x, state[pfx + "ln_2.weight"], state[pfx + "ln_2.bias"])
    h = quick_gelu(F.linear(h, state[pfx + "mlp.c_fc.weight"], state[pfx + "mlp.c_fc.bias"]))
    h = F.linear(h, state[pfx + "mlp.c_proj.weight"], state[pfx + "mlp.c_proj.bias"])
    return x + h


def count_blocks(state: State, pfx: str) -> int:
    n = 0
    while f"{pfx}{n}.ln_1.weight" in state:
        n += 1
    return n


def clip_vision_tokens(state: State, img: torch.Tensor, pfx: str = "clip_encoder.visual."):
    """VisionTransformer.forward_with_embeddings (models/clip/model.py:296-317):
    patchify conv (no bias) -> prepend class token -> + pos -> ln_pre -> blocks ->
    ln_post on ALL tokens, no output projection.  [B,3,S,S] -> [B,g*g+1,W]."""
    wconv = state[pfx + "conv1.weight"]
    width, _, p, _ = wconv.shape
    x = F.conv2d(img, wconv, bias=None, stride=p)           # [B,W,g,g]
    x = x.flatten(2).transpose(1, 2)                        # [B,g*g,W] row-major patches
    cls = state[pfx + "class_embedding"].expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1) + state[pfx + "positional_embedding"]
    x = layer_norm(x, state[pfx + "ln_pre.weight"], state[pfx + "ln_pre.bias"])
    heads = width // 64                                      # models/clip/model.py:351
    for i in range(count_blocks(state, pfx + "transformer.resblocks.")):
        x = clip_resblock(state, f"{pfx}transformer.resblocks.{i}.", x, heads, causal=False)
    return layer_norm(x, state[pfx + "ln_post.weight"], state[pfx + "ln_post.bias"])


def clip_text_tokens(state: State, tokens: torch.Tensor, pfx: str = "clip_encoder."):
    """CLIP.encode_text_with_embeddings (models/clip/model.py:447-461):
    embedding gather + pos -> causal blocks -> ln_final; no EOT pooling,
    no text_projection.  [B,77] int -> [B,77,W]."""
    emb = state[pfx + "token_embedding.weight"]
    x = emb[tokens.long()] + state[pfx + "positional_embedding"]
    heads = emb.shape[1] // 64                               # models/clip/model.py:532
    for i in range(count_blocks(state, pfx + "transformer.resblocks.")):
        x = clip_resblock(state, f"{pfx}transformer.resblocks.{i}.", x, heads, causal=True)
    return layer_norm(x, state[pfx + "ln_final.weight"], state[pfx + "ln_final.bias"])


# --------------------------------------------------------------------------
# SigLIP towers (third-party arithmetic: `transformers` SiglipModel, unpinned in the
# reference's pyproject.toml:31; restated from its published definition and pinned by
# running transformers 5.15 itself in oracle/gen_golden.py)
# --------------------------------------------------------------------------
def gelu_tanh(x: torch.Tensor):
    """`gelu_pytorch_tanh`: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))."""
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x * x * x)))


def siglip_layer(state: State, pfx: str, x: torch.Tensor, heads: int, eps: float = 1e-6):
    """SiglipEncoderLayer: pre-LN, q/k/v/out projections WITH bias, scale d^-0.5, no mask,
    MLP fc1 -> tanh-GELU -> fc2."""
    h = layer_norm(x, state[pfx + "layer_norm1.weight"], state[pfx + "layer_norm1.bias"], eps)
    q, k, v = (split_heads(F.linear(h, state[f"{pfx}self_attn.{n}_proj.weight"],
                                    state[f"{pfx}self_attn.{n}_proj.bias"]), heads) for n in "qkv")
    o = attention_core(q, k, v, scale=(x.shape[-1] // heads) ** -0.5)
    x = x + F.linear(merge_heads(o), state[pfx + "self_attn.out_proj.weight"],
                     state[pfx + "self_attn.out_proj.bias"])
    h = layer_norm(x, state[pfx + "layer_norm2.weight"], state[pfx + "layer_norm2.bias"], eps)
    h = gelu_tanh(F.linear(h, state[pfx + "mlp.fc1.weight"], state[pfx + "mlp.fc1.bias"]))
    return x + F.linear(h, state[pfx + "mlp.fc2.weight"], state[pfx + "mlp.fc2.bias"])


def _siglip_encoder(state: State, pfx: str, x: torch.Tensor, heads: int):
    i = 0
    while f"{pfx}encoder.layers.{i}.layer_norm1.weight" in state:
        x = siglip_layer(state, f"{pfx}encoder.layers.{i}.", x, heads)
        i += 1
    return x


def siglip_vision_tokens(state: State, img: torch.Tensor, heads: int,
                         pfx: str = "siglip_model.vision_model."):
    """vision_model(pixel_values).last_hidden_state: patch conv (with bias) + learned position
    embedding (no class token) -> encoder -> post_layernorm.  [B,3,S,S] -> [B,(S/p)^2,W].
    (The attention-pooling head is computed by HF and discarded by the reference,
    models/siglip.py:59-60.)"""
    w = state[pfx + "embeddings.patch_embedding.weight"]
    x = F.conv2d(img, w, state[pfx + "embeddings.patch_embedding.bias"], stride=w.shape[-1])
    x = x.flatten(2).transpose(1, 2) + state[pfx + "embeddings.position_embedding.weight"]
    x = _siglip_encoder(state, pfx, x, heads)
    return layer_norm(x, state[pfx + "post_layernorm.weight"], state[pfx + "post_layernorm.bias"], 1e-6)


def siglip_text_tokens(state: State, ids: torch.Tensor, heads: int,
                       pfx: str = "siglip_model.text_model."):
    """text_model(input_ids).last_hidden_state: token + position embeddings -> encoder (no
    attention mask) -> final_layer_norm.  [B,L] int -> [B,L,W]."""
    x = state[pfx + "embeddings.token_embedding.weight"][ids.long()]
    x = x + state[pfx + "embeddings.position_embedding.weight"][: ids.shape[1]]
    x = _siglip_encoder(state, pfx, x, heads)
    return layer_norm(x, state[pfx + "final_layer_norm.weight"], state[pfx + "final_layer_norm.bias"], 1e-6)


# --------------------------------------------------------------------------
# trainable tail: fusion transformer + heatmap decoders
# --------------------------------------------------------------------------
def fusion_layer(state: State, pfx: str, x: torch.Tensor, heads: int,
                 query_row_mask: Optional[torch.Tensor] = None, drop=None):
    """One Transformer layer (models/transformer.py:44-64, modules.py:8-30):
    x = Attn(LN x) + x ; x = FF(LN x) + x.  to_qkv has NO bias, scale = d_h^-0.5,
    FF = Linear -> erf-GELU -> Linear.  `drop` = (p, {"attn", "out", "act", "ff": keep masks}) applies the block's four
    nn.Dropout sites (transformer.py:18,20-24,36; modules.py:24,26) with GIVEN masks; None = the shipped p = 0."""
    dp = (lambda n: (drop[0], drop[1][n])) if drop is not None else (lambda n: None)
    w = x.shape[-1]
    h = layer_norm(x, state[pfx + "0.norm.weight"], state[pfx + "0.norm.bias"])
    qkv = F.linear(h, state[pfx + "0.fn.to_qkv.weight"])
    q, k, v = (split_heads(t, heads) for t in qkv.chunk(3, dim=-1))
    o = attention_core(q, k, v, scale=(w // heads) ** -0.5, query_row_mask=query_row_mask, p_drop=dp("attn"))
    o = _drop(F.linear(merge_heads(o), state[pfx + "0.fn.to_out.0.weight"],
                       state[pfx + "0.fn.to_out.0.bias"]), dp("out"))
    x = o + x
    h = layer_norm(x, state[pfx + "1.norm.weight"], state[pfx + "1.norm.bias"])
    h = _drop(erf_gelu(F.linear(h, state[pfx + "1.fn.net.0.weight"], state[pfx + "1.fn.net.0.bias"])), dp("act"))
    h = _drop(F.linear(h, state[pfx + "1.fn.net.3.weight"], state[pfx + "1.fn.net.3.bias"]), dp("ff"))
    return h + x


def concat_transformer(state: State, inputs: Sequence[torch.Tensor], heads: int,
                       modalities: Optional[Sequence[int]] = None,
                       attention_masks: Optional[torch.Tensor] = None,
                       pfx: str = "pick_place.fusion.", layer_drop=None):
    """ConcatTransformer.forward (models/fusion.py:35-64): add the modality's
    type embedding to each input, concatenate along the sequence, run the
    pre-norm transformer (no final norm), return the last input's tokens."""
    if modalities is None:
        modalities = range(len(inputs))
    type_emb = state[pfx + "token_type_embeddings.weight"]
    x = torch.cat([inp + type_emb[m] for m, inp in zip(modalities, inputs)], dim=1)
    depth = 0
    while f"{pfx}transformer_encoder.layers.{depth}.0.norm.weight" in state:
        x = fusion_layer(state, f"{pfx}transformer_encoder.layers.{depth}.", x, heads,
                         attention_masks, drop=None if layer_drop is None else layer_drop[depth])
        depth += 1
    return x[:, -inputs[-1].shape[1]:, :]


def conv_decoder_logits(state: State, pfx: str, feat: torch.Tensor):
    """ConvDecoder as written (models/modules.py:33-74): five 1x1 convs (with
    bias, NO activation) interleaved with four bilinear x2 upsamples
    (align_corners=False).  [B,D,g,g] -> [B,1,16g,16g]."""
    x = feat
    for i in (0, 2, 4, 6, 8):
        x = F.conv2d(x, state[f"{pfx}decoder_net.{i}.weight"], state[f"{pfx}decoder_net.{i}.bias"])
        if i != 8:
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    return x


def decoder_heads(is_bimanual: bool):
    """Head order of PickPlaceConvDecoder (models/pickplace.py:34-47)."""
    if is_bimanual:
        return ["left_pick", "right_pick", "left_place", "right_place"]
    return ["pick", "place"]


def pick_place_convdecoder(state: State, inputs: Sequence[torch.Tensor], heads: int,
                           is_bimanual: bool, pfx: str = "pick_place.", compute_mask: Optional[bool] = None,
                           detach_mask: bool = False, **fusion_kw):
    """PickPlaceConvDecoder.forward (models/pickplace.py:59-124): fuse, drop token 0, view tokens as a row-major
    g x g grid -> [B,D,g,g], one ConvDecoder per head, squeeze, sigmoid.

    compute_mask (default: whether the state holds a `mask_head`; models/pickplace.py:25-30,66-69,76-98,107-119):
    a further ConvDecoder reads the LAST input's tokens (minus token 0) -- as the fusion model left them, i.e. WITH the
    type embedding its forward added in place (models/fusion.py:52: `inp += type_embeddings`) -- and its sigmoid is
    `mask_heatmap`; every pick heatmap is multiplied by it (by its detached value with detach_mask)."""
    if compute_mask is None:
        compute_mask = f"{pfx}mask_head.decoder_net.0.weight" in state
    fused = concat_transformer(state, inputs, heads, pfx=pfx + "fusion.", **fusion_kw)
    tok = fused[:, 1:, :]
    b, n, d = tok.shape
    g = int(math.isqrt(n))
    feat = tok.reshape(b, g, g, d).permute(0, 3, 1, 2)
    out: Dict[str, Optional[torch.Tensor]] = {"attn_weights": None}
    mask = None
    if compute_mask:
        mods = fusion_kw.get("modalities")
        last_mod = list(mods)[-1] if mods is not None else len(inputs) - 1
        raw = inputs[-1] + state[pfx + "fusion.token_type_embeddings.weight"][last_mod]
        mfeat = raw[:, 1:, :].reshape(b, g, g, d).permute(0, 3, 1, 2)
        mask = torch.sigmoid(conv_decoder_logits(state, f"{pfx}mask_head.", mfeat).squeeze(1))
        out["mask_heatmap"] = mask
    for name in decoder_heads(is_bimanual):
        logits = conv_decoder_logits(state, f"{pfx}{name}_decoder.", feat)
        heat = torch.sigmoid(logits.squeeze(1))
        if mask is not None and name.endswith("pick"):
            heat = heat * (mask.detach() if detach_mask else mask)
        out[f"{name}_heatmap"] = heat
    return out


def upsample_operator(g: int, levels: int = 4, dtype=torch.float64) -> torch.Tensor:
    """The fixed [g*2^levels, g] matrix U such that `levels` successive bilinear x2
    upsamples (align_corners=False) of a g-vector v equal U @ v.  Used by the
    collapsed form of the decoder (the decoder is affine: SURVEY.md K9)."""
    eye = torch.eye(g, dtype=dtype).reshape(g, 1, g)         # g signals of length g
    x = eye
    for _ in range(levels):
        x = F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
    return x.reshape(g, -1).t().contiguous()                 # [S, g]


def collapse_decoder(state: State, pfx: str):
    """Fold the five affine 1x1 convs into (w_eff [D], b_eff scalar).  Bilinear rows
    sum to 1, so biases commute with the upsamples exactly (in real arithmetic)."""
    w_eff = None
    b_eff = None
    for i in (0, 2, 4, 6, 8):
        w = state[f"{pfx}decoder_net.{i}.weight"].flatten(1).double()   # [Cout,Cin]
        b = state[f"{pfx}decoder_net.{i}.bias"].double()
        if w_eff is None:
            w_eff, b_eff = w, b
        else:
            w_eff, b_eff = w @ w_eff, w @ b_eff + b
    return w_eff.reshape(-1), b_eff.reshape(())


# --------------------------------------------------------------------------
# model, loss, optimiser
# --------------------------------------------------------------------------
def rgb_clip_forward(state: State, sample: Mapping[str, torch.Tensor], heads: int,
                     is_bimanual: bool = False, dropout=None, detach_mask: bool = False):
    """RGBOnly.forward (models/rgb_clip.py:52-73).  `dropout` (training with p > 0, GIVEN masks):
    dict(text=(p, keep [B,78,D]), rgb=(p, keep [B,n+1,D]), layers=[(p, {attn, out, act, ff})...])."""
    t = clip_text_tokens(state, sample["instruction"])
    b = t.shape[0]
    t = torch.cat([state["text_token"].expand(b, 1, -1), t], dim=1)
    dr = dropout or {}
    x_text = _drop(t + state["text_pos_embedding"][:, : t.shape[1]], dr.get("text"))
    v = clip_vision_tokens(state, sample["rgb"])
    x_rgb = _drop(F.linear(v, state["project.weight"], state["project.bias"]) + state["rgb_pos_embedding"], dr.get("rgb"))
    kw = dict(layer_drop=dr["layers"]) if dr.get("layers") is not None else {}
    return pick_place_convdecoder(state, [x_text, x_rgb], heads, is_bimanual, detach_mask=detach_mask, **kw)


def _prepend(token: torch.Tensor, feats: torch.Tensor):
    return torch.cat([token.expand(feats.shape[0], 1, -1), feats], dim=1)


def siglip_forward(state: State, sample: Mapping[str, torch.Tensor], tower_heads: int, fusion_heads: int,
                   is_bimanual: bool, detach_mask: bool = False):
    """SigLip.forward (models/siglip.py:56-70): prepend the learned image / text tokens to the
    towers' last hidden states and hand both to the pick/place head."""
    img = _prepend(state["image_token"], siglip_vision_tokens(state, sample["rgb"], tower_heads))
    txt = _prepend(state["text_token"], siglip_text_tokens(state, sample["instruction"], tower_heads))
    return pick_place_convdecoder(state, [txt, img], fusion_heads, is_bimanual, detach_mask=detach_mask)


def siglip_sequential_forward(state: State, sample: Mapping[str, torch.Tensor], tower_heads: int,
                              fusion_heads: int, is_bimanual: bool, detach_mask: bool = False):
    """SiglipSequential.forward (models/siglip.py:85-137): the vision tower also encodes the T
    context frames; each gets the image token prepended, frames are concatenated and
    `context_pos_embedding` added; the fusion mask is [ones | ctx_mask repeated n+1 | ones] and
    masks QUERY rows; modalities = [0, 1, 1]."""
    img = _prepend(state["image_token"], siglip_vision_tokens(state, sample["rgb"], tower_heads))
    txt = _prepend(state["text_token"], siglip_text_tokens(state, sample["instruction"], tower_heads))
    ctx = sample["rgb_context"]
    b, t = ctx.shape[:2]
    n = img.shape[1]
    cf = siglip_vision_tokens(state, ctx.flatten(0, 1), tower_heads)              # [(b t), n-1, W]
    cf = _prepend(state["image_token"], cf).reshape(b, t * n, -1) + state["context_pos_embedding"]
    mask = torch.cat([torch.ones(b, txt.shape[1]),
                      sample["context_attention_mask"].float().repeat_interleave(n, dim=1),
                      torch.ones(b, n)], dim=1)
    return pick_place_convdecoder(state, [txt, cf, img], fusion_heads, is_bimanual, detach_mask=detach_mask,
                                  modalities=[0, 1, 1], attention_masks=mask)


def bce_mean(p: torch.Tensor, t: torch.Tensor):
    """nn.BCELoss(reduction='mean') incl. PyTorch's clamp of each log at -100."""
    lp = torch.log(p).clamp_min(-100.0)
    l1p = torch.log(1.0 - p).clamp_min(-100.0)
    return -(t * lp + (1.0 - t) * l1p).mean()


def bce_gaussmap(output: Mapping[str, torch.Tensor], sample: Mapping[str, torch.Tensor],
                 is_bimanual: bool, mask_pick_heatmap: bool = False):
    """BCEGaussMap (losses/__init__.py:54-109): per-head mean BCE, summed over heads;
    optional target*mask on pick heads."""
    names = ([f"{a}_{k}" for a in ("left", "right") for k in ("pick", "place")]
             if is_bimanual else ["pick", "place"])
    parts = {}
    total = None
    for n in names:
        tgt = sample[f"{n}_heatmap"]
        if mask_pick_heatmap and n.endswith("pick"):
            tgt = tgt * sample["mask"].squeeze(1)
        cur = bce_mean(output[f"{n}_heatmap"], tgt)
        parts[n] = cur
        total = cur if total is None else total + cur
    return total, parts


def bce_mask(output: Mapping[str, torch.Tensor], sample: Mapping[str, torch.Tensor]):
    """BCEMask (losses/__init__.py:112-117): mean BCE of the mask head against sample["mask"] [B,1,S,S]."""
    return bce_mean(output["mask_heatmap"], sample["mask"].squeeze(1)), {}


def dice_loss(output: Mapping[str, torch.Tensor], sample: Mapping[str, torch.Tensor]):
    """DiceLoss (losses/__init__.py:126-136): per sample 1 - (2 sum(p t) + 1) / (sum p + sum t + 1), summed."""
    p = output["mask_heatmap"].flatten(1)
    t = sample["mask"].flatten(1)
    num = 2.0 * (p * t).sum(1)
    den = p.sum(-1) + t.sum(-1)
    return (1.0 - (num + 1.0) / (den + 1.0)).sum(), {}


def focal_loss(output: Mapping[str, torch.Tensor], sample: Mapping[str, torch.Tensor], alpha: float = 0.25,
               gamma: float = 2):
    """FocalLoss (losses/__init__.py:139-155): alpha_t * BCE * (1 - p_t)^gamma per pixel; `.mean(1).sum()` on
    [B,S,S] = the sum over everything divided by the number of ROWS."""
    p = output["mask_heatmap"]
    t = sample["mask"].squeeze(1)
    ce = F.binary_cross_entropy(p, t, reduction="none")      # the reference's call: its backward is torch's own formula
    p_t = p * t + (1.0 - p) * (1.0 - t)
    loss = ce * ((1.0 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * t + (1.0 - alpha) * (1.0 - t)) * loss
    return loss.mean(1).sum(), {}


def composed_loss(output, sample, loss_names: Sequence[str], weights: Sequence[float], is_bimanual: bool,
                  mask_pick_heatmap: bool = False, **kw):
    """ComposedLoss (losses/__init__.py:30-51): the weighted sum of named losses; the parts dict carries each loss
    under its name and its own parts under "name part"."""
    fns = {"bce_gaussmap": lambda o, s: bce_gaussmap(o, s, is_bimanual, mask_pick_heatmap),
           "bce_mask": bce_mask, "dice": dice_loss,
           "focal": lambda o, s: focal_loss(o, s, **{k: v for k, v in kw.items() if k in ("alpha", "gamma")})}
    total, parts = None, {}
    for n, w in zip(loss_names, weights):
        cur, sub = fns[n](output, sample)
        total = cur * w if total is None else total + cur * w
        parts[n] = cur
        for k, v in sub.items():
            parts[f"{n} {k}"] = v
    return total, parts


def adam_update(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8,
                weight_decay=0.0, decoupled=False):
    """torch.optim.Adam / AdamW single-tensor math (optim/__init__.py:4-14 selects
    them; formula = PyTorch's _single_tensor_adam, `step` is 1-based).  Returns
    new (p, m, v)."""
    if weight_decay != 0.0:
        if decoupled:
            p = p * (1.0 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def linear_warmup_lrs(base_lr: float, max_iters: int, n_steps: int, warmup_portion=0.1,
                      warmup_start_lr=0.0, use_cosine_decay=True):
    """LR seen by optimizer.step() number 0,1,2,... under LinearWarmup
    (optim/__init__.py:28-92).  Quirk kept: last_step starts at -1 and step() runs
    in __init__, so the first LR is start + (base-start)*(-1/warmup) (negative for
    start 0); warm-up runs while last_step <= warmup_steps, then CosineAnnealingLR
    (T_max = max_iters - warmup_steps, eta_min 0) takes over from base."""
    warm = int(warmup_portion * max_iters)
    lrs = []
    last = -1
    lr = base_lr
    cos_t = 0
    t_max = max_iters - warm

    def sched_step():
        nonlocal last, lr, cos_t
        if last <= warm:
            lr = warmup_start_lr + (base_lr - warmup_start_lr) * (last / warm)
            last += 1
        elif use_cosine_decay:
            # closed form of CosineAnnealingLR's recursion, eta_min = 0
            cos_t += 1
            lr = base_lr * (1.0 + math.cos(math.pi * cos_t / t_max)) / 2.0

    sched_step()                       # the call in __init__
    for _ in range(n_steps):
        lrs.append(lr)                 # LR used by this optimizer.step()
        sched_step()
    return lrs


# --------------------------------------------------------------------------
# parameter construction (same keys / shapes / init family as the reference)
# --------------------------------------------------------------------------
def make_clip_state(gen: torch.Generator, *, image_resolution=224, vision_layers=12,
                    vision_width=768, vision_patch_size=16, context_length=77,
                    vocab_size=49408, transformer_width=512, transformer_layers=12,
                    embed_dim=512, pfx="clip_encoder.") -> Dict[str, torch.Tensor]:
    """Random CLIP weights with the reference's key names and init scales
    (models/clip/model.py:250-262, 367-412).  Values come from this generator,
    not from the reference's RNG stream: fixtures store the weights explicitly."""
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std

    s: Dict[str, torch.Tensor] = {}
    vw, tw = vision_width, transformer_width
    g = image_resolution // vision_patch_size
    v = pfx + "visual."
    s[v + "conv1.weight"] = rn(vw, 3, vision_patch_size, vision_patch_size,
                               std=(3 * vision_patch_size ** 2) ** -0.5)
    s[v + "class_embedding"] = rn(vw, std=vw ** -0.5)
    s[v + "positional_embedding"] = rn(g * g + 1, vw, std=vw ** -0.5)
    s[v + "proj"] = rn(vw, embed_dim, std=vw ** -0.5)
    for ln in ("ln_pre", "ln_post"):
        s[v + ln + ".weight"] = 1.0 + 0.1 * rn(vw)
        s[v + ln + ".bias"] = 0.1 * rn(vw)

    def blocks(prefix, width, layers):
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        for i in range(layers):
            b = f"{prefix}transformer.resblocks.{i}."
            s[b + "attn.in_proj_weight"] = rn(3 * width, width, std=width ** -0.5)
            s[b + "attn.in_proj_bias"] = 0.02 * rn(3 * width)
            s[b + "attn.out_proj.weight"] = rn(width, width, std=proj_std)
            s[b + "attn.out_proj.bias"] = 0.02 * rn(width)
            for ln in ("ln_1", "ln_2"):
                s[b + ln + ".weight"] = 1.0 + 0.1 * rn(width)
                s[b + ln + ".bias"] = 0.1 * rn(width)
            s[b + "mlp.c_fc.weight"] = rn(4 * width, width, std=(2 * width) ** -0.5)
            s[b + "mlp.c_fc.bias"] = 0.02 * rn(4 * width)
            s[b + "mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std)
            s[b + "mlp.c_proj.bias"] = 0.02 * rn(width)

    blocks(v, vw, vision_layers)
    blocks(pfx, tw, transformer_layers)
    s[pfx + "token_embedding.weight"] = rn(vocab_size, tw, std=0.02)
    s[pfx + "positional_embedding"] = rn(context_length, tw, std=0.01)
    s[pfx + "ln_final.weight"] = 1.0 + 0.1 * rn(tw)
    s[pfx + "ln_final.bias"] = 0.1 * rn(tw)
    s[pfx + "text_projection"] = rn(tw, embed_dim, std=tw ** -0.5)
    s[pfx + "logit_scale"] = torch.tensor(math.log(1 / 0.07))
    return s


def make_tail_state(gen: torch.Generator, *, dim=512, vision_width=768, n_text=78, n_rgb=197,
                    depth=8, mlp_ratio=4, is_bimanual=False, num_modalities=2
                    ) -> Dict[str, torch.Tensor]:
    """Random trainable-tail weights, reference keys (models/rgb_clip.py:34-50,
    fusion.py:8-33, transformer.py:8-58, modules.py:33-74)."""
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std

    def lin(prefix, out_f, in_f, bias=True):
        bound = in_f ** -0.5
        s[prefix + ".weight"] = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
        if bias:
            s[prefix + ".bias"] = (torch.rand(out_f, generator=gen) * 2 - 1) * bound

    s: Dict[str, torch.Tensor] = {}
    s["text_token"] = rn(1, 1, dim)
    s["text_pos_embedding"] = rn(1, n_text, dim)
    s["rgb_pos_embedding"] = rn(1, n_rgb, dim)
    lin("project", dim, vision_width)
    f = "pick_place.fusion."
    s[f + "token_type_embeddings.weight"] = rn(num_modalities, dim, std=0.02)
    for i in range(depth):
        b = f"{f}transformer_encoder.layers.{i}."
        for n in ("0.norm", "1.norm"):
            s[b + n + ".weight"] = 1.0 + 0.1 * rn(dim)
            s[b + n + ".bias"] = 0.1 * rn(dim)
        lin(b + "0.fn.to_qkv", 3 * dim, dim, bias=False)
        lin(b + "0.fn.to_out.0", dim, dim)
        lin(b + "1.fn.net.0", dim * mlp_ratio, dim)
        lin(b + "1.fn.net.3", dim, dim * mlp_ratio)
    chans = [dim, dim // 2, dim // 2, dim // 4, dim // 4, 1]
    for name in decoder_heads(is_bimanual):
        for j, i in enumerate((0, 2, 4, 6, 8)):
            cin, cout = chans[j], chans[j + 1]
            bound = cin ** -0.5
            s[f"pick_place.{name}_decoder.decoder_net.{i}.weight"] = (
                (torch.rand(cout, cin, 1, 1, generator=gen) * 2 - 1) * bound)
            s[f"pick_place.{name}_decoder.decoder_net.{i}.bias"] = (
                (torch.rand(cout, generator=gen) * 2 - 1) * bound)
    return s


def is_trainable_key(key: str) -> bool:
    """RGBOnly freezes every parameter whose name contains 'clip_encoder'
    (models/rgb_clip.py:31, models/__init__.py:70-73)."""
    return "clip_encoder" not in key


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def gaussian_heatmaps(centres: torch.Tensor, size: int, sigma: float = 5.0):
    """One peak-1.0 Gaussian per map; centres [B,2] as (x, y) pixels."""
    ys = torch.arange(size, dtype=torch.float32).view(1, size, 1)
    xs = torch.arange(size, dtype=torch.float32).view(1, 1, size)
    cx = centres[:, 0].float().view(-1, 1, 1)
    cy = centres[:, 1].float().view(-1, 1, 1)
    return torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2.0 * sigma ** 2))


def synthetic_sample(batch: int, gen: torch.Generator, *, size=224, context=77, vocab=49408,
                     real_tokens=32, is_bimanual=False, heat_size=None) -> Dict[str, torch.Tensor]:
    rgb = torch.randn(batch, 3, size, size, generator=gen)
    instr = torch.zeros(batch, context, dtype=torch.int32)
    instr[:, 0] = vocab - 2                                   # <|startoftext|> 49406
    body = torch.randint(1, vocab - 2, (batch, real_tokens - 2), generator=gen)
    instr[:, 1:real_tokens - 1] = body.int()
    instr[:, real_tokens - 1] = vocab - 1                     # <|endoftext|> 49407
    sample = {"rgb": rgb, "instruction": instr}
    for n in decoder_heads(is_bimanual):
        # the decoder always upsamples the patch grid x16 (modules.py:68-69): with a patch size other
        # than 16 (ViT-L/14) the heatmaps are 16 * grid pixels wide, not the image size
        hs = heat_size or size
        c = torch.randint(0, hs, (batch, 2), generator=gen)
        sample[f"{n}_heatmap"] = gaussian_heatmaps(c, hs)
    return sample


# ---------------------------------------------------------------------------------------
# Either side of the step: heatmap -> action, pixel -> target heatmap.
# numpy like the reference; pinned by tests/golden/postproc.npz (oracle/gen_golden.py).
# ---------------------------------------------------------------------------------------
def nearest_to_mask(u, v, mask):
    """(row, col) of the non-zero mask pixel closest to (u, v): first minimum of the squared
    distance over np.argwhere's row-major order (bifold/models/utils.py:80-83)."""
    idx = np.argwhere(mask)
    return idx[((idx - [u, v]) ** 2).sum(1).argmin()]


def sample_from_heatmap(heatmap, mask=None):
    """heatmap [B,S,S], mask [B,1,S,S] or None -> (pixels int64 [B,2] as (x, y), confidences
    fp32 [B]) (bifold/models/utils.py:60-77)."""
    hm = np.asarray(heatmap, dtype=np.float32)
    pixels, confidences = [], []
    for i in range(hm.shape[0]):
        r, c = np.unravel_index(hm[i].argmax(), hm[i].shape)
        if mask is not None:
            r, c = nearest_to_mask(r, c, np.asarray(mask[i]).squeeze())
        pixels.append([c, r])
        confidences.append(hm[i][r, c])
    return np.array(pixels, dtype=np.int64), np.array(confidences, dtype=np.float32)


def select_action(output, mask, is_bimanual, threshold=0.01, constrain_pick_mask=True):
    """Heatmaps -> {name: int64 [B,2]} pixel action (bifold/models/__init__.py:79-173, image
    branch).  Bimanual: an arm is kept if its pick confidence reaches the threshold or it is the
    more confident arm; a dropped arm's pick and place are (-1, -1)."""
    if not is_bimanual:
        pick, _ = sample_from_heatmap(output["pick_heatmap"], mask if constrain_pick_mask else None)
        place, _ = sample_from_heatmap(output["place_heatmap"])
        return {"pick": pick, "place": place}
    lp, lc = sample_from_heatmap(output["left_pick_heatmap"], mask)
    rp, rc = sample_from_heatmap(output["right_pick_heatmap"], mask)
    pick = np.stack((lp, rp))
    conf = np.stack((lc, rc))
    b = conf.shape[-1]
    keep = np.logical_or(conf >= threshold, conf.argmax(axis=0) == np.tile(np.arange(2), (b, 1)).T)
    pick[~keep] = -1
    lplace, _ = sample_from_heatmap(output["left_place_heatmap"])
    rplace, _ = sample_from_heatmap(output["right_place_heatmap"])
    lplace[~keep[0]] = -1
    rplace[~keep[1]] = -1
    return {"left_pick": pick[0], "right_pick": pick[1], "left_place": lplace, "right_place": rplace}


def make_gaussmap(points, size: int, sigma: float = 5.0, strategy: str = "gmm"):
    """Target heatmap of one head (bifold/data/processor.py:405-439).  points: None (all-zero
    map) or fp32 [P,2] (x, y) pixels; one point, or strategy `first`: peak-1 Gaussian at the
    rounded (half-to-even) first point; `gmm`: sum of normalised Gaussians / its maximum; `fit`: the
    density of a bivariate normal fitted to the (unrounded) points.
    fp64 arithmetic, fp32 result."""
    grid = np.arange(size)
    x, y = np.meshgrid(grid, grid)
    if points is None:
        return np.zeros((size, size), dtype=np.float32)
    points = np.asarray(points, dtype=np.float32)
    if len(points) == 1 or strategy == "first":
        cx, cy = round(points[0, 0]), round(points[0, 1])
        g = np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sigma * sigma))
    elif strategy == "gmm":
        g = np.zeros((size, size), dtype=np.float64)
        for cx, cy in np.round(points):
            d = (x - np.float64(cx)) ** 2 + (y - np.float64(cy)) ** 2
            g += np.exp(-d / (2 * sigma ** 2)) / ((2 * np.pi) * sigma ** 2)
        g /= g.max()
    elif strategy == "fit":
        # scipy.stats.multivariate_normal(mean, cov).pdf restated in closed form (processor.py:427-436): mean and the
        # ddof = 1 covariance of the UNROUNDED points, density not normalised; a singular covariance raises in scipy
        # numpy semantics of the reference's two calls: np.mean of the fp32 points is an fp32 (sequential) sum / n,
        # np.cov promotes to fp64 and centres with ITS OWN fp64 mean
        mean = np.mean(points, axis=0).astype(np.float64)
        cov = np.cov(points, rowvar=False)
        det = cov[0, 0] * cov[1, 1] - cov[0, 1] * cov[1, 0]
        if not det > 0:
            raise np.linalg.LinAlgError("singular covariance: the `fit` strategy needs three non-collinear points")
        inv = np.array([[cov[1, 1], -cov[0, 1]], [-cov[1, 0], cov[0, 0]]]) / det
        dx, dy = x - mean[0], y - mean[1]
        maha = dx * dx * inv[0, 0] + 2 * dx * dy * inv[0, 1] + dy * dy * inv[1, 1]
        g = np.exp(-0.5 * maha) / (2 * np.pi * np.sqrt(det))
    else:
        raise ValueError(f"Strategy {strategy} not recognized")
    return g.astype(np.float32)


def preprocess_rgb(raw, mask, mean, std):
    """uint8 [S,S,3] (+ 0/1 mask [S,S]) -> fp32 [3,S,S]: mask fill 77 (bifold/data/processor.py:255-258), then
    rgb_transform (:87-95) for a frame already at the model's resolution, where v2.Resize / CenterCrop are the
    identity: ToDtype(float32, scale=True) = x / 255, Normalize = (x - mean) / std.  torchvision is absent in
    this container, so this formula is restated from the transform definitions (parity unpinned for it)."""
    raw = np.asarray(raw)
    if mask is not None:
        m = np.asarray(mask)
        raw = (raw * m[:, :, None] + (1 - m[:, :, None]) * 77).astype(np.uint8)
    x = raw.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    mean = np.asarray(mean, dtype=np.float32)[:, None, None]
    std = np.asarray(std, dtype=np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32)


def resize_normalize_rgb(raw, mask, size: int, mean, std):
    """The reference's rgb_transform for frames of any resolution (bifold/data/processor.py:45-50, 87-95, 255-258):
    mask fill 77 -> v2.ToImage -> v2.Resize(size, BICUBIC, antialias) -> v2.CenterCrop(size) ->
    v2.ToDtype(float32, scale=True) -> v2.Normalize.  raw uint8 [B, H, W, 3].  PARITY UNPINNED: torchvision is
    absent, so its documented pipeline is restated over torch's own uint8 antialiased bicubic
    F.interpolate -- the kernel torchvision dispatches to for uint8 CPU images.  Second witness (round 3): PIL's
    antialiased BICUBIC, which the reference itself calls for `raw_rgb` (processor.py:263), agrees to one grey level
    (two below a 2x shrink) on < 1 % of the pixels: tests/test_processor.py."""
    x = torch.as_tensor(raw)
    if mask is not None:
        m = torch.as_tensor(mask).reshape(x.shape[0], x.shape[1], x.shape[2], 1).float()
        x = torch.floor(x.float() * m + (1.0 - m) * 77.0).to(torch.uint8)
    b, h, w, _ = x.shape
    new_h, new_w = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)
    img = F.interpolate(x.permute(0, 3, 1, 2).contiguous(), size=(new_h, new_w), mode="bicubic", antialias=True,
                        align_corners=False)
    top, left = int(round((new_h - size) / 2.0)), int(round((new_w - size) / 2.0))
    img = img[:, :, top:top + size, left:left + size].float() / 255.0
    mean = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return (img - mean) / std


def depth_frames(depth, mask=None, shift=None, standardize: bool = False, thresh: float = 0.1):
    """The depth transform chain of the reference's Processor for frames at the model's own resolution
    (bifold/data/processor.py:59-78, where torchvision's Resize / CenterCrop are the identity): DepthScale --
    `depth + shift`, one draw per frame (data/depth_augmentations.py:35-43) -- MaskDepth -- `depth * mask`
    (:46-51) -- float32, and TruncatedDepthStandardization (:6-17): with the frame's values sorted, drop
    int(thresh * n) from the bottom and keep up to int((1 - thresh) * n), subtract the mean of that slice and divide by
    sqrt(its unbiased variance + 1e-6).  depth [B, S, S] float32.  PINNED: tests/golden/depth.npz is a run of the
    reference's own classes (oracle/gen_processor_golden.py)."""
    d = np.asarray(depth, dtype=np.float32)
    if shift is not None:
        d = d + np.asarray(shift, dtype=np.float32).reshape(-1, 1, 1)
    if mask is not None:
        d = d * np.asarray(mask, dtype=np.float32)
    d = d.astype(np.float32)
    if not standardize:
        return d
    out = np.empty_like(d)
    for b in range(d.shape[0]):
        frame = torch.from_numpy(d[b])
        srt = torch.sort(frame.reshape(-1), dim=0)[0]
        n = srt.shape[0]
        srt = srt[int(thresh * n):int((1 - thresh) * n)]
        out[b] = ((frame - srt.mean()) / torch.sqrt(srt.var() + 1e-6)).numpy()
    return out


def affine_nearest(x, angle: float, dx: float, dy: float):
    """v2.functional.affine(x, angle, translate=[dx, dy], scale=1.0, shear=[0.0]) with its defaults (nearest, zero
    fill, centre of the image) as Processor._spatial_augmentation calls it (processor.py:333-341).  PARITY UNPINNED
    (torchvision absent): its tensor path restated -- inverse matrix [cos, sin, tx'; -sin, cos, ty'] about the
    centre, a base grid of pixel centres, normalisation by half the size, grid_sample(nearest, zeros,
    align_corners=False)."""
    import math
    x = torch.as_tensor(x).float()
    n, c, h, w = x.shape
    rot = math.radians(angle)
    m = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]
    m[2] += m[0] * (-dx) + m[1] * (-dy)
    m[5] += m[3] * (-dx) + m[4] * (-dy)
    theta = torch.tensor(m, dtype=torch.float32).view(1, 2, 3)
    xs = torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, w)
    ys = torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, h)
    base = torch.empty(1, h, w, 3)
    base[..., 0] = xs.view(1, 1, w)
    base[..., 1] = ys.view(1, h, 1)
    base[..., 2] = 1.0
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h])
    grid = (base.view(1, h * w, 3) @ rescaled).view(1, h, w, 2).expand(n, h, w, 2)
    return F.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)

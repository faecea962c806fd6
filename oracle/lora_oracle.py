"""CPU restatement of the LoRA-adapted SigLIP towers of SigLip / SiglipSequential with `lora: true`
(reference bifold/models/siglip.py:33-40: `LoraModel(siglip_model, LoraConfig(r, lora_alpha,
target_modules, lora_dropout), "siglip_adapter")`).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the product package never imports this.

PINNED AT DROPOUT 0, UNPINNED FOR THE DROPOUT MASK.  The adapter arithmetic lives in the third-party package `peft`
(unpinned in the reference's pyproject.toml:28), which is not installed in this image and is neither imported nor
stood in for.  Since round 3 the file is pinned by a run of the reference itself wherever dropout is the identity
(p = 0, eval mode): oracle/gen_golden.py::siglip_lora_fixture runs the reference's `lora=False` model around towers
whose q_proj / v_proj weights are W' = W + (alpha / r) B A, records heatmaps, loss, tail gradients and dL/dW', and
asserts this file's unmerged forward and autograd gradients against them (adapter gradients through the chain rule
dA = (alpha / r) B^T G, dB = (alpha / r) G A^T); tests/test_host_logic.py repeats it from the committed fixtures
(tests/golden/tiny_siglip*_lora.npz).  The one thing assumed about PEFT is that defining formula.  What stays
unpinned is training-mode `lora_dropout` > 0: nn.Dropout on the adapter branch's input, replayed here from the
masks the implementation under test drew.  This
file restates PEFT's published definition of a LoRA-wrapped nn.Linear

        y = base_layer(x) + lora_B(lora_A(lora_dropout(x))) * (lora_alpha / r)

(peft.tuners.lora.layer.Linear.forward; lora_A: Linear(in, r, bias=False) initialised
kaiming_uniform(a=sqrt(5)), lora_B: Linear(r, out, bias=False) initialised to zero; lora_dropout is
nn.Dropout(p) on the adapter branch's input -- the identity at p = 0 / in eval mode, and with a
replayed mask here, see lora_linear) and PEFT's parameter naming after wrapping
(`base_model.model.<hf name>`, `<proj>.base_layer.{weight,bias}`,
`<proj>.lora_{A,B}.<adapter>.weight`).  It deliberately applies the adapter UNMERGED, as PEFT does in
training, so that it is an independent check of the product's merged-weight formulation; gradients
come from torch autograd.  Everything around the adapters reuses oracle/bifold_oracle.py (pinned
against the reference by tests/golden/tiny_siglip*.npz).
"""
from __future__ import annotations

from typing import Mapping, Sequence

import torch
import torch.nn.functional as F

from . import bifold_oracle as O

State = Mapping[str, torch.Tensor]
PEFT_PREFIX = "siglip_model.base_model.model."


def lora_linear(state: State, pfx: str, x: torch.Tensor, scale: float, adapter: str):
    """`pfx` = "...self_attn.q_proj." -- adapted when the state holds lora_A for it.

    Training-mode dropout (`lora_dropout` > 0): nn.Dropout's mask is random, so a checker must REPLAY the
    mask the implementation under test drew.  `state["__dropout__"]` (optional) = (p, {pfx: [mask, ...]}):
    the keep masks (1 = kept, shaped like x) of that Linear in call order; each call consumes one and
    applies inverted dropout, x * mask / (1 - p), to the adapter branch only -- the base branch sees x."""
    a_key = f"{pfx}lora_A.{adapter}.weight"
    if a_key not in state:
        return F.linear(x, state[pfx + "weight"], state[pfx + "bias"])
    y = F.linear(x, state[pfx + "base_layer.weight"], state[pfx + "base_layer.bias"])
    xa = x
    if "__dropout__" in state:
        p, masks = state["__dropout__"]
        xa = x * masks[pfx].pop(0).to(x.dtype).reshape(x.shape) / (1.0 - p)
    return y + F.linear(F.linear(xa, state[a_key]), state[f"{pfx}lora_B.{adapter}.weight"]) * scale


def siglip_layer(state: State, pfx: str, x: torch.Tensor, heads: int, scale: float, adapter: str,
                 eps: float = 1e-6):
    """bifold_oracle.siglip_layer with PEFT-wrapped q/k/v projections."""
    h = O.layer_norm(x, state[pfx + "layer_norm1.weight"], state[pfx + "layer_norm1.bias"], eps)
    q, k, v = (O.split_heads(lora_linear(state, f"{pfx}self_attn.{n}_proj.", h, scale, adapter), heads)
               for n in "qkv")
    o = O.attention_core(q, k, v, scale=(x.shape[-1] // heads) ** -0.5)
    x = x + F.linear(O.merge_heads(o), state[pfx + "self_attn.out_proj.weight"],
                     state[pfx + "self_attn.out_proj.bias"])
    h = O.layer_norm(x, state[pfx + "layer_norm2.weight"], state[pfx + "layer_norm2.bias"], eps)
    h = O.gelu_tanh(F.linear(h, state[pfx + "mlp.fc1.weight"], state[pfx + "mlp.fc1.bias"]))
    return x + F.linear(h, state[pfx + "mlp.fc2.weight"], state[pfx + "mlp.fc2.bias"])


def _encoder(state: State, pfx: str, x: torch.Tensor, heads: int, scale: float, adapter: str):
    i = 0
    while f"{pfx}encoder.layers.{i}.layer_norm1.weight" in state:
        x = siglip_layer(state, f"{pfx}encoder.layers.{i}.", x, heads, scale, adapter)
        i += 1
    return x


def vision_tokens(state: State, img: torch.Tensor, heads: int, scale: float, adapter: str,
                  pfx: str = PEFT_PREFIX + "vision_model."):
    w = state[pfx + "embeddings.patch_embedding.weight"]
    x = F.conv2d(img, w, state[pfx + "embeddings.patch_embedding.bias"], stride=w.shape[-1])
    x = x.flatten(2).transpose(1, 2) + state[pfx + "embeddings.position_embedding.weight"]
    x = _encoder(state, pfx, x, heads, scale, adapter)
    return O.layer_norm(x, state[pfx + "post_layernorm.weight"], state[pfx + "post_layernorm.bias"], 1e-6)


def text_tokens(state: State, ids: torch.Tensor, heads: int, scale: float, adapter: str,
                pfx: str = PEFT_PREFIX + "text_model."):
    x = state[pfx + "embeddings.token_embedding.weight"][ids.long()]
    x = x + state[pfx + "embeddings.position_embedding.weight"][: ids.shape[1]]
    x = _encoder(state, pfx, x, heads, scale, adapter)
    return O.layer_norm(x, state[pfx + "final_layer_norm.weight"], state[pfx + "final_layer_norm.bias"], 1e-6)


def siglip_forward(state: State, sample, tower_heads: int, fusion_heads: int, is_bimanual: bool,
                   r: int, lora_alpha: float, adapter: str = "siglip_adapter"):
    """SigLip.forward (models/siglip.py:56-70) over LoRA-wrapped towers."""
    s = lora_alpha / r
    img = O._prepend(state["image_token"], vision_tokens(state, sample["rgb"], tower_heads, s, adapter))
    txt = O._prepend(state["text_token"], text_tokens(state, sample["instruction"], tower_heads, s, adapter))
    return O.pick_place_convdecoder(state, [txt, img], fusion_heads, is_bimanual)


def siglip_sequential_forward(state: State, sample, tower_heads: int, fusion_heads: int, is_bimanual: bool,
                              r: int, lora_alpha: float, adapter: str = "siglip_adapter"):
    """SiglipSequential.forward (models/siglip.py:85-137) over LoRA-wrapped towers: the vision tower is
    applied to the current image and, separately, to the T context frames (two calls, as the reference)."""
    s = lora_alpha / r
    img = O._prepend(state["image_token"], vision_tokens(state, sample["rgb"], tower_heads, s, adapter))
    txt = O._prepend(state["text_token"], text_tokens(state, sample["instruction"], tower_heads, s, adapter))
    ctx = sample["rgb_context"]
    b, t = ctx.shape[:2]
    n = img.shape[1]
    cf = vision_tokens(state, ctx.flatten(0, 1), tower_heads, s, adapter)
    cf = O._prepend(state["image_token"], cf).reshape(b, t * n, -1) + state["context_pos_embedding"]
    mask = torch.cat([torch.ones(b, txt.shape[1]),
                      sample["context_attention_mask"].float().repeat_interleave(n, dim=1),
                      torch.ones(b, n)], dim=1)
    return O.pick_place_convdecoder(state, [txt, cf, img], fusion_heads, is_bimanual,
                                    modalities=[0, 1, 1], attention_masks=mask)


def trainable_keys(state: State) -> Sequence[str]:
    """PEFT freezes everything inside the wrapped model except the adapters; the rest of SigLip
    (tokens, context positions, fusion, decoders) trains as before."""
    return [k for k in state if k != "__dropout__" and (not k.startswith("siglip_model.") or ".lora_" in k)]

"""Frozen SigLIP image/text towers on libbifold_hip.

The reference obtains them from `transformers.AutoModel.from_pretrained(automodel_name)`
(bifold/models/siglip.py:5,26) and uses only `vision_model(...).last_hidden_state` and
`text_model(...).last_hidden_state` (:59-60,65,102-104): the pooled heads and logits that HF
also computes are discarded, so they are not computed here.  Parameter names are HF's
(`vision_model.embeddings.patch_embedding.weight`, `text_model.encoder.layers.{i}.self_attn.q_proj.weight`
...; full list pinned by tests/golden/tiny_siglip.npz), so a SiglipModel state dict loads with
`load_state_dict`.  Block = pre-LN (eps 1e-6), q/k/v/out projections with bias, tanh-GELU MLP; no
class token, no attention mask.

`lora=None`: frozen towers, forward only (models/siglip.py:41-43).  With a `lora` dict (the
reference's default, models/siglip.py:33-40: `LoraModel(siglip_model, LoraConfig(r, lora_alpha,
target_modules=[q_proj, v_proj], lora_dropout), "siglip_adapter")`) every targeted projection of
BOTH towers gains `lora_A [r, in]` (Kaiming-uniform) and `lora_B [out, r]` (zeros), activations are
kept, and `backward()` walks both towers in reverse to produce the adapters' gradients.  Two forms
of PEFT's `Wx + b + (alpha / r) B A dropout(x)`:
  * merged -- `W + (alpha / r) B A` folded into the packed in-projection: exact when the dropout is
    the identity, i.e. at lora_dropout = 0 and, as in PEFT, whenever the module is in eval mode
    or gradients are off;
  * unmerged (training with lora_dropout > 0, the reference's shipped 0.01): the base projection
    plus, per adapted projection, `u = dropout(h) A^T` (an [M, r] GEMM) and a rank-r GEMM that
    ACCUMULATES `(alpha / r) u B^T` into that projection's columns of the packed QKV buffer.  Every
    adapted Linear draws its own mask, regenerated from a seed in the backward pass
    (bf_dropout); `last_dropout_seeds` lets a checker replay the masks.  Parameter names then follow PEFT's
wrapping (`base_model.model.` prefix, `<proj>.base_layer.{weight,bias}`,
`<proj>.lora_{A,B}.<adapter>.weight`).  PARITY UNPINNED for this mode: `peft` is not in the
image, so the arithmetic follows PEFT's published definition and is checked against a torch
autograd restatement of it (oracle/lora_oracle.py), not against the reference itself.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Dict, List, Optional

import torch
from torch import nn

from bifold_amd.hip import lib as L
from bifold_amd.hip import ops
from .arena import ParamArena
from .params import add_param as _register
from .params import get_param, layernorm_init, linear_init

_BASE = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12)
# Hyper-parameters of the published checkpoints the reference's configs name
# (conf/model/siglip*.yaml: google/siglip-base-patch16-{image_size}).
SIGLIP_CONFIGS: Dict[str, Dict] = {
    f"google/siglip-base-patch16-{s}": dict(vision=dict(_BASE, image_size=s, patch_size=16),
                                            text=dict(_BASE, vocab_size=32000, max_position_embeddings=64))
    for s in (224, 256, 384, 512)
}
LN_EPS = 1e-6
_TOWER_GRAD_STREAM16 = True       # the LoRA towers' backward in fp16 mode: residual-stream gradient as fp16 only (see _encoder_bwd)


_MERGEABLE = ("q_proj", "k_proj", "v_proj")      # projections folded into the packed in-projection


class SiglipTowers(nn.Module):
    def __init__(self, vision: Dict, text: Dict, lora: Optional[Dict] = None):
        super().__init__()
        self.vcfg, self.tcfg = dict(vision), dict(text)
        self.lora = None
        self._real: Dict[str, str] = {}               # HF parameter name -> registered path
        self._prefix = ""
        if lora:
            targets = tuple(lora["targets"])
            bad = [t for t in targets if t not in _MERGEABLE]
            if bad:
                raise NotImplementedError(f"LoRA target_modules {bad}: only {list(_MERGEABLE)} are built")
            if int(lora["r"]) % 4 != 0 or int(lora["r"]) <= 0:
                raise ValueError(f"LoRA rank r={lora['r']} must be a positive multiple of 4 (fp32 GEMM row alignment)")
            p_drop = float(lora.get("dropout", 0.0))
            if not 0.0 <= p_drop < 1.0:
                raise ValueError(f"lora_dropout must be in [0, 1), got {p_drop}")
            recompute = str(lora.get("recompute", "none"))
            if recompute not in ("none", "mlp", "full"):
                raise ValueError(f"lora_recompute must be none, mlp or full, got {recompute!r}")
            self.lora = dict(r=int(lora["r"]), alpha=float(lora["alpha"]), targets=targets,
                             adapter=str(lora.get("adapter", "siglip_adapter")), dropout=p_drop, recompute=recompute)
            self._prefix = "base_model.model."
        vw, tw = vision["hidden_size"], text["hidden_size"]
        p = vision["patch_size"]
        self.grid = vision["image_size"] // p
        self.patch = p
        n = self.grid ** 2
        for k, v in linear_init(vw, 3 * p * p).items():
            self._add(f"vision_model.embeddings.patch_embedding.{k}", v.view(vw, 3, p, p) if k == "weight" else v)
        self._add("vision_model.embeddings.position_embedding.weight", torch.randn(n, vw) * vw ** -0.5)
        self._layers("vision_model", vision)
        for k, v in layernorm_init(vw).items():
            self._add(f"vision_model.post_layernorm.{k}", v)
        # attention-pooling head: part of the checkpoint schema, never evaluated (its output is unused)
        self._add("vision_model.head.probe", torch.randn(1, 1, vw))
        self._add("vision_model.head.attention.in_proj_weight", torch.randn(3 * vw, vw) * vw ** -0.5)
        self._add("vision_model.head.attention.in_proj_bias", torch.zeros(3 * vw))
        for k, v in linear_init(vw, vw).items():
            self._add(f"vision_model.head.attention.out_proj.{k}", v)
        for k, v in layernorm_init(vw).items():
            self._add(f"vision_model.head.layernorm.{k}", v)
        for k, v in linear_init(vision["intermediate_size"], vw).items():
            self._add(f"vision_model.head.mlp.fc1.{k}", v)
        for k, v in linear_init(vw, vision["intermediate_size"]).items():
            self._add(f"vision_model.head.mlp.fc2.{k}", v)
        self._add("text_model.embeddings.token_embedding.weight", torch.randn(text["vocab_size"], tw) * tw ** -0.5)
        self._add("text_model.embeddings.position_embedding.weight",
                  torch.randn(text["max_position_embeddings"], tw) * tw ** -0.5)
        self._layers("text_model", text)
        for k, v in layernorm_init(tw).items():
            self._add(f"text_model.final_layer_norm.{k}", v)
        for k, v in linear_init(tw, tw).items():
            self._add(f"text_model.head.{k}", v)
        self._add("logit_scale", torch.zeros(1))
        self._add("logit_bias", torch.zeros(1))
        self._arena: Optional[ParamArena] = None
        self._packed: Dict[str, torch.Tensor] = {}
        self._packed_version = -1
        self._err_flag: Optional[torch.Tensor] = None
        self._saved: Dict[str, Dict] = {}             # per tower: what backward() needs
        self._packed_merged = False
        self._unmerged_now = False
        self._lora_packed = {}                    # tower -> operands of the bf_lora_* kernels (None: GEMM form)
        # (tower, call index within the step, layer, projection) -> seed of that Linear's dropout mask in the
        # most recent training forward; bf_dropout_mask(seed) replays it
        self.last_dropout_seeds: Dict = {}
        self._calls: Dict[str, int] = {}
        if self.lora:                                 # PEFT: only the adapters train
            for n, p in self.named_parameters():
                p.requires_grad_(".lora_" in n)

    def _add(self, name: str, value: torch.Tensor) -> nn.Parameter:
        """Register the parameter HF calls `name` under the path the checkpoint schema uses."""
        real = self._prefix + name
        if self.lora:
            for t in self.lora["targets"]:
                tag = f".self_attn.{t}."
                if tag in name:
                    real = self._prefix + name.replace(tag, tag + "base_layer.")
        self._real[name] = real
        return _register(self, real, value)

    def _p(self, name: str) -> nn.Parameter:
        return get_param(self, self._real[name])

    def lora_params(self, tower: str, layer: int, proj: str):
        """(lora_A [r, in], lora_B [out, r]) of one adapted projection."""
        base = f"{self._prefix}{tower}.encoder.layers.{layer}.self_attn.{proj}."
        ad = self.lora["adapter"]
        return get_param(self, f"{base}lora_A.{ad}.weight"), get_param(self, f"{base}lora_B.{ad}.weight")

    def _layers(self, tower: str, cfg: Dict):
        w, inter = cfg["hidden_size"], cfg["intermediate_size"]
        for i in range(cfg["num_hidden_layers"]):
            b = f"{tower}.encoder.layers.{i}."
            for ln in ("layer_norm1", "layer_norm2"):
                for k, v in layernorm_init(w).items():
                    self._add(f"{b}{ln}.{k}", v)
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):          # HF's registration order
                for k, v in linear_init(w, w).items():
                    self._add(f"{b}self_attn.{proj}.{k}", v)
                if self.lora and proj in self.lora["targets"]:
                    # peft LoraLayer.reset_lora_parameters: A ~ kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(in)), B = 0
                    r, ad = self.lora["r"], self.lora["adapter"]
                    base = f"{self._prefix}{b}self_attn.{proj}."
                    bound = 1.0 / math.sqrt(w)
                    _register(self, f"{base}lora_A.{ad}.weight", torch.empty(r, w).uniform_(-bound, bound))
                    _register(self, f"{base}lora_B.{ad}.weight", torch.zeros(w, r))
            for k, v in linear_init(inter, w).items():
                self._add(f"{b}mlp.fc1.{k}", v)
            for k, v in linear_init(w, inter).items():
                self._add(f"{b}mlp.fc2.{k}", v)

    # ------------------------------------------------------------------ runtime
    def _runtime(self, cdt: torch.dtype) -> ParamArena:
        a = self._arena
        if a is None or not a.still_bound() or (cdt != torch.float32 and a.shadow is None):
            # the adapters live in the MODEL's trainable arena (gradients, Adam state), not here
            a = ParamArena([(n, p) for n, p in self.named_parameters() if ".lora_" not in n],
                           with_grad=False, with_shadow=(cdt != torch.float32),
                           shadow_dtype=cdt if cdt != torch.float32 else torch.bfloat16)
            self._arena = a
            self._packed_version = -1
        if cdt != torch.float32:
            a.set_shadow_dtype(cdt)
        a.sync_shadow()
        ver = sum(p._version for p in a.params)
        if ver != self._packed_version or self._packed.get("dtype") != cdt:
            # q/k/v live as three parameters (checkpoint schema) but run as ONE in-projection GEMM
            # that writes the packed [B, L, 3, H, D] buffer the attention kernels read
            self._packed = {"dtype": cdt}
            for tower, cfg in (("vision_model", self.vcfg), ("text_model", self.tcfg)):
                for i in range(cfg["num_hidden_layers"]):
                    b = f"{tower}.encoder.layers.{i}.self_attn."
                    ws = [a.operand(self._p(f"{b}{n}_proj.weight"), cdt) for n in "qkv"]
                    bs = [self._p(f"{b}{n}_proj.bias").data for n in "qkv"]
                    self._packed[b + "w"] = torch.cat(ws, dim=0).contiguous()
                    self._packed[b + "b"] = torch.cat(bs, dim=0).contiguous()
            self._packed_version = ver
            self._packed_merged = False
        if self.lora and not self._unmerged():
            self._merge_adapters(cdt)
            self._packed_merged = True
        elif self._packed_merged:                         # back to the base rows for the unmerged form
            self._packed_version = -1
            return self._runtime(cdt)
        return a

    def _unmerged(self) -> bool:
        """PEFT's training-mode forward with an active dropout: the adapters cannot be folded into W.
        Decided by _keep() at the top of a tower call (the body runs under no_grad)."""
        return self._unmerged_now

    def begin_step(self) -> None:
        """Called by the model at the top of each forward: resets the per-step call counters that key the
        dropout seeds (the vision tower runs twice per step in SiglipSequential)."""
        self._calls = {}
        self.last_dropout_seeds = {}

    def _merge_adapters(self, cdt: torch.dtype) -> None:
        """Packed in-projection rows of every adapted projection <- W + (alpha / r) B A (fp32 math on
        the fp32 masters, then one rounding to the compute dtype).  Runs every forward: the adapters
        change every optimiser step, the base weights never."""
        scale = self.lora["alpha"] / self.lora["r"]
        for tower, cfg in (("vision_model", self.vcfg), ("text_model", self.tcfg)):
            w = cfg["hidden_size"]
            for i in range(cfg["num_hidden_layers"]):
                b = f"{tower}.encoder.layers.{i}.self_attn."
                pw = self._packed[b + "w"]
                for slot, n in enumerate("qkv"):
                    if f"{n}_proj" not in self.lora["targets"]:
                        continue
                    la, lb = self.lora_params(tower, i, f"{n}_proj")
                    rows = pw[slot * w:(slot + 1) * w]
                    base = self._p(f"{b}{n}_proj.weight").data
                    if cdt == torch.float32:
                        ops.gemm(lb.data, la.data, b_kn=True, alpha=scale, residual=base, out=rows)
                    else:
                        _, c16 = ops.gemm(lb.data, la.data, b_kn=True, alpha=scale, residual=base,
                                          want_bf16_copy=True)
                        rows.copy_(c16)

    def _w(self, name: str, cdt):
        return self._arena.operand(self._p(name), cdt)

    def _f(self, name: str):
        return self._p(name).data

    def _encoder_frozen(self, x, tower: str, cfg: Dict, batch: int, seq: int, cdt, final: str):
        """The pre-LN blocks of a FROZEN tower in a 16-bit mode, followed by the tower's last LayerNorm `final` (fp32
        output).  The form of the CLIP towers (clip_encoder._tower): out-proj and fc2 write their branch as a 16-bit
        tensor (plain epilogue, no fp32 residual tile), the residual adds ride in the LayerNorm passes, and because
        nothing reads the stream between a block's two branches it is written once per block
        (bf_add2_layernorm_fwd): ln_2(x + attn) without an x_out, then ln_1'((x + attn) + mlp) with one."""
        heads, w = cfg["num_attention_heads"], cfg["hidden_size"]
        d = w // heads
        n_layers = cfg["num_hidden_layers"]
        h1 = None
        for i in range(n_layers):
            b = f"{tower}.encoder.layers.{i}."
            if h1 is None:
                h1, _, _ = self._ln1(x, b, cdt, False)
            qkv, _ = self._qkv(h1, b, i, tower, cdt, w, None)
            o, _ = ops.attention_fwd(qkv, batch, seq, heads, d, d ** -0.5, L.MASK_NONE, need_lse=False)
            attn = ops.gemm(o, self._w(b + "self_attn.out_proj.weight", cdt), bias=self._f(b + "self_attn.out_proj.bias"))
            h2, _, _, _ = ops.add_layernorm_fwd(x, attn, self._f(b + "layer_norm2.weight"), self._f(b + "layer_norm2.bias"),
                                                cdt, eps=LN_EPS, write_x=False)
            a = ops.gemm(h2, self._w(b + "mlp.fc1.weight", cdt), bias=self._f(b + "mlp.fc1.bias"), act=L.ACT_TANH_GELU)
            mlp = ops.gemm(a, self._w(b + "mlp.fc2.weight", cdt), bias=self._f(b + "mlp.fc2.bias"))
            if i + 1 < n_layers:
                nb = f"{tower}.encoder.layers.{i + 1}."
                h1, _, _, _ = ops.add_layernorm_fwd(x, attn, self._f(nb + "layer_norm1.weight"),
                                                    self._f(nb + "layer_norm1.bias"), cdt, eps=LN_EPS, delta2=mlp)
            else:
                y, _, _, _ = ops.add_layernorm_fwd(x, attn, self._f(final + ".weight"), self._f(final + ".bias"),
                                                   torch.float32, eps=LN_EPS, delta2=mlp, write_x=False)
                return y
        raise RuntimeError("a tower without layers")

    def _frozen_fast(self, keep: bool, cfg: Dict, cdt) -> bool:
        # OPT-IN (BIFOLD_SIGLIP_LAZY=1).  Measured at configs[2] (B = 128, fp16, same box, two runs each): 279.4 / 279.9 ->
        # 281.2 / 282.7 samples/s (+0.8 %), and the logits of the timed batch against the oracle 7.6e-4 -> 8.5e-4 of
        # north_star's 1e-3 (full-size test: 7.9e-4 -> 8.1e-4): the branches are rounded to 16 bits before the fp32 add,
        # 48 times per tower, where the epilogue form adds the fp32 accumulator.  This path has no parity to spend.
        return (not keep and cdt in ops.HALF and cfg["hidden_size"] in (256, 512, 768, 1024)
                and cfg["num_hidden_layers"] > 0 and os.environ.get("BIFOLD_SIGLIP_LAZY", "0") == "1")

    def _encoder(self, x, tower: str, cfg: Dict, batch: int, seq: int, cdt, saved: Optional[List] = None):
        """The pre-LN blocks.  `saved` (a list) receives per layer what _encoder_bwd needs; without it
        the residual stream is updated in place."""
        heads = cfg["num_attention_heads"]
        w = cfg["hidden_size"]
        d = w // heads
        keep = saved is not None
        unmerged = keep and self._unmerged()
        if unmerged:
            if cdt != torch.float32 and self.lora["r"] % 8:
                raise ValueError(f"lora_dropout > 0 in bf16 needs a LoRA rank that is a multiple of 8 (the adapter "
                                 f"branch is a K = r bf16 GEMM), got r={self.lora['r']}")
            call = self._calls.get(tower, 0)
            self._calls[tower] = call + 1
            # one host-side draw per tower call from torch's CPU generator (seeded by cfg.seed): no device sync
            base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
            # the Trainer seeds (and on resume restores) that generator identically on every data-parallel rank:
            # fold the rank in, so replicas drop different elements like one process over the whole batch would
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                base_seed ^= (torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
            self._lora_pack(tower, cfg, cdt)
        policy = self.lora["recompute"] if keep else "none"
        for i in range(cfg["num_hidden_layers"]):
            b = f"{tower}.encoder.layers.{i}."
            seeds = None
            if unmerged:
                seeds = {}
                for slot, n in enumerate("qkv"):
                    if f"{n}_proj" in self.lora["targets"]:
                        seeds[n] = (base_seed + 16 * i + slot) & (2 ** 63 - 1)
                        self.last_dropout_seeds[(tower, call, i, f"{n}_proj")] = seeds[n]
            h1, m1, r1 = self._ln1(x, b, cdt, keep)
            qkv, drop = self._qkv(h1, b, i, tower, cdt, w, seeds)
            o, lse = ops.attention_fwd(qkv, batch, seq, heads, d, d ** -0.5, L.MASK_NONE, need_lse=keep)
            xm = ops.gemm(o, self._w(b + "self_attn.out_proj.weight", cdt), bias=self._f(b + "self_attn.out_proj.bias"),
                          residual=x, out=None if keep else x, out_dtype=torch.float32)
            h2, m2, r2 = ops.layernorm_fwd(xm, self._f(b + "layer_norm2.weight"), self._f(b + "layer_norm2.bias"), cdt,
                                           save_stats=keep, eps=LN_EPS)
            if keep:
                a, pre = ops.gemm(h2, self._w(b + "mlp.fc1.weight", cdt), bias=self._f(b + "mlp.fc1.bias"),
                                  act=L.ACT_TANH_GELU, want_pre=True)
            else:
                a, pre = ops.gemm(h2, self._w(b + "mlp.fc1.weight", cdt), bias=self._f(b + "mlp.fc1.bias"),
                                  act=L.ACT_TANH_GELU), None
            xo = ops.gemm(a, self._w(b + "mlp.fc2.weight", cdt), bias=self._f(b + "mlp.fc2.bias"), residual=xm,
                          out=None if keep else xm, out_dtype=torch.float32)
            if keep:      # h2 and a feed only frozen weights' gradients: never kept
                S = dict(x=x, m1=m1, r1=r1, h1=h1, qkv=qkv, o=o, lse=lse, xm=xm, m2=m2, r2=r2, pre=pre, drop=drop,
                         seeds=seeds)
                # Activation recomputation (round 4; reference models/siglip.py:85-137 keeps the whole autograd graph of
                # both towers for every frame -- at B = 128 x 5 frames x 576 tokens x 12 layers that is 88 GB of the
                # step's 0.2 TB).  "mlp": h1 and the MLP's pre-activation (10 of 26 bytes x width per token and layer)
                # are re-made in the backward from x / xm -- one LayerNorm each and the fc1 GEMM, 30 % of a block's
                # forward FLOPs; "full": only the block input x stays, the backward re-runs the block up to fc1
                # (70 %).  The re-made tensors are bit for bit the forward's (same kernels, same inputs, the same
                # dropout seeds), so gradients do not depend on the policy.  Measured at configs[2], B = 128, fp16
                # (tools/lora_recompute_bench.py, five runs in one process each): none 198 samples/s at 181 GiB peak,
                # mlp 190 at 147 GiB, full 179 at 109 GiB -- all stable since the dW side stream stopped pinning its
                # operands with record_stream (pickplace._WeightGradSide).  Default: none (288 GB is there to be used).
                if policy in ("mlp", "full"):
                    S["h1"] = S["pre"] = None
                if policy == "full":
                    S["qkv"] = S["o"] = S["lse"] = S["xm"] = S["m2"] = S["r2"] = S["drop"] = None
                saved.append(S)
            x = xo
        return x

    def _ln1(self, x, b: str, cdt, keep: bool):
        return ops.layernorm_fwd(x, self._f(b + "layer_norm1.weight"), self._f(b + "layer_norm1.bias"), cdt,
                                 save_stats=keep, eps=LN_EPS)

    def _qkv(self, h1, b: str, i: int, tower: str, cdt, w: int, seeds):
        """Packed QKV projection; with `seeds` (training with lora_dropout > 0) the adapters run unmerged behind their own
        dropout masks: qkv[:, slot] += s (D h1) A^T B^T.  Returns (qkv, {proj letter: (u, seed)} or None)."""
        pk = self._lora_packed.get(tower) if seeds is not None else None
        if pk is not None:
            # the adapter branch of every adapted projection (csrc/lora.hip): u = (D . h1) A^T / (1 - p) from ONE read of
            # h1, then  qkv[:, slot] += scaling u B^T  -- as a low-rank addend INSIDE the projection's GEMM (one more MFMA
            # K-step per output tile, bf_gemm_args::lr_u) where the persistent kernel takes the call, else as a
            # read-modify-write pass over the q / v column blocks (bf_lora_up: the text tower's few rows)
            p_drop = self.lora["dropout"]
            sd = [seeds[n] for n in pk["names"]]
            nsr = len(sd) * pk["r"]
            ride = pk["bcat"] is not None and ops.gemm_lowrank_ok(h1.shape[0], 3 * w, w, cdt, nsr)
            if ride:
                u, u16 = ops.lora_down(h1, [0] * len(sd), pk["a"][i], pk["r"], w, p_drop, sd, alpha=1.0 / (1.0 - p_drop),
                                       want_half=True)
                qkv = ops.gemm(h1, self._packed[b + "self_attn.w"], bias=self._packed[b + "self_attn.b"], lr_u=u16,
                               lr_w=pk["bcat"][i])
            else:
                qkv = ops.gemm(h1, self._packed[b + "self_attn.w"], bias=self._packed[b + "self_attn.b"])
                u = ops.lora_down(h1, [0] * len(sd), pk["a"][i], pk["r"], w, p_drop, sd, alpha=1.0 / (1.0 - p_drop))
                ops.lora_up(u, pk["bt"][i], qkv, pk["offs"], pk["r"], w)
            return qkv, dict(u=u, seeds=sd)
        qkv = ops.gemm(h1, self._packed[b + "self_attn.w"], bias=self._packed[b + "self_attn.b"])
        if seeds is None:
            return qkv, None
        p_drop, scale = self.lora["dropout"], self.lora["alpha"] / self.lora["r"]
        drop = {}
        for slot, n in enumerate("qkv"):
            if n not in seeds:
                continue
            la, lb = self.lora_params(tower, i, f"{n}_proj")
            hd = ops.dropout(h1, p_drop, seeds[n])
            u = ops.gemm(hd, la.data.to(cdt))                          # [M, r]
            ops.gemm(u, lb.data.to(cdt), alpha=scale, out=qkv[:, slot * w:(slot + 1) * w], accumulate=True)
            drop[n] = (u, seeds[n])
            del hd
        return qkv, drop

    def _lora_pack(self, tower: str, cfg: Dict, cdt) -> None:
        """Operands of the bf_lora_* kernels for one tower call: every layer's lora_A rows stacked ([layers, S r, width])
        and scaling * lora_B^T in the same layout, in the compute dtype -- three launches per tower call instead of a cast
        per use.  None for shapes the kernels do not take (fp32 mode, other ranks): those run dropout + GEMMs."""
        w, r = cfg["hidden_size"], self.lora["r"]
        names = [n for n in "qkv" if f"{n}_proj" in self.lora["targets"]]
        self._lora_packed[tower] = None
        # BIFOLD_LORA_FUSED: 1 (default) these kernels, the up-projection riding in the GEMM where it can; "up": these
        # kernels with the up-projection always as its own pass; 0: the dropout + GEMM form of rounds 3 - 5
        if os.environ.get("BIFOLD_LORA_FUSED", "1") == "0" or not ops.lora_supported(cdt, w, r, len(names)):
            return
        scale = self.lora["alpha"] / r
        n_layers = cfg["num_hidden_layers"]
        pairs = [self.lora_params(tower, i, f"{n}_proj") for i in range(n_layers) for n in names]
        a = torch.cat([la.data for la, _ in pairs]).to(cdt).view(n_layers, len(names) * r, w)
        bt = (torch.cat([lb.data.t() for _, lb in pairs]) * scale).to(cdt).view(n_layers, len(names) * r, w)
        offs = ["qkv".index(n) * w for n in names]
        bcat = None
        if len(names) * r in (16, 32) and os.environ.get("BIFOLD_LORA_FUSED", "1") != "up":
            # scaling * lora_B of every adapted projection in the rows (of the packed projection) and columns (of u) that
            # are its own, zero elsewhere: the lr_w operand of the projection's GEMM
            bcat = torch.zeros(n_layers, 3 * w, len(names) * r, dtype=cdt, device=a.device)
            for s_, off in enumerate(offs):
                bcat[:, off:off + w, s_ * r:(s_ + 1) * r] = bt[:, s_ * r:(s_ + 1) * r, :].transpose(1, 2)
        self._lora_packed[tower] = dict(a=a, bt=bt, bcat=bcat, names=names, r=r, offs=offs)

    def _rematerialize(self, S: Dict, b: str, i: int, tower: str, cfg: Dict, batch: int, seq: int, cdt) -> Dict:
        """Fill in what the forward dropped under the recompute policy, by re-running its kernels on what it kept."""
        heads, w = cfg["num_attention_heads"], cfg["hidden_size"]
        d = w // heads
        if S["h1"] is None:
            S["h1"], m1, r1 = self._ln1(S["x"], b, cdt, True)
            if S["m1"] is None:
                S["m1"], S["r1"] = m1, r1
        if S["qkv"] is None:
            S["qkv"], S["drop"] = self._qkv(S["h1"], b, i, tower, cdt, w, S["seeds"])
        if S["o"] is None:
            S["o"], S["lse"] = ops.attention_fwd(S["qkv"], batch, seq, heads, d, d ** -0.5, L.MASK_NONE, need_lse=True)
        if S["xm"] is None:
            S["xm"] = ops.gemm(S["o"], self._w(b + "self_attn.out_proj.weight", cdt),
                               bias=self._f(b + "self_attn.out_proj.bias"), residual=S["x"], out_dtype=torch.float32)
        if S["pre"] is None:
            h2, m2, r2 = ops.layernorm_fwd(S["xm"], self._f(b + "layer_norm2.weight"), self._f(b + "layer_norm2.bias"),
                                           cdt, save_stats=True, eps=LN_EPS)
            if S["m2"] is None:
                S["m2"], S["r2"] = m2, r2
            S["pre"] = ops.gemm(h2, self._w(b + "mlp.fc1.weight", cdt), bias=self._f(b + "mlp.fc1.bias"))
        return S

    def _encoder_bwd(self, dx: torch.Tensor, dx_c: Optional[torch.Tensor], saved: List, tower: str, cfg: Dict,
                     batch: int, seq: int, cdt, grad_view) -> None:
        """dx fp32 [M, W] (+ its compute-dtype copy) = d loss / d (last block's output).  Walks the blocks in reverse; the only
        parameter gradients produced are the adapters' (everything else is frozen):
        with W_eff = W + s B A,   dB = s dW_eff A^T,   dA = s B^T dW_eff,   dW_eff = dY^T h1."""
        heads = cfg["num_attention_heads"]
        w = cfg["hidden_size"]
        d = w // heads
        scale = self.lora["alpha"] / self.lora["r"]
        dev = dx.device
        junk_g = torch.empty(w, dtype=torch.float32, device=dev)       # LayerNorm weight/bias gradients: frozen
        junk_b = torch.empty(w, dtype=torch.float32, device=dev)
        dw = torch.empty(3 * w, w, dtype=torch.float32, device=dev)
        # fp16 mode: the gradient of the residual stream travels between the blocks' LayerNorm backward passes as fp16 only
        # (the fusion stack's _GRAD_STREAM16, pickplace.py: every add rounds the running gradient to 11 bits, the loss scale
        # keeps it in range): a pass reads 2 + 4 + 2 and writes 2 bytes per element instead of 2 + 4 + 4 and 4 + 2 --
        # 561 -> 350 us per pass at the vision tower's 368 640 x 768, 48 passes per step.  bf16 (8 bits) and fp32 keep fp32.
        s16 = cdt == torch.float16 and dx_c is not None and _TOWER_GRAD_STREAM16
        if s16:
            dx = None
        for i in reversed(range(cfg["num_hidden_layers"])):
            b = f"{tower}.encoder.layers.{i}."
            S = self._rematerialize(saved[i], b, i, tower, cfg, batch, seq, cdt)
            op = dx_c if dx_c is not None else dx
            # MLP:  xo = xm + gelu_tanh(h2 W1^T + b1) W2^T + b2
            dpre = ops.gemm(op, self._w(b + "mlp.fc2.weight", cdt), b_kn=True, act=L.ACT_TANH_GELU_GRAD, aux=S["pre"])
            dh2 = ops.gemm(dpre, self._w(b + "mlp.fc1.weight", cdt), b_kn=True)
            dxm, dxm_c = ops.layernorm_bwd(dh2, S["xm"], self._f(b + "layer_norm2.weight"), S["m2"], S["r2"],
                                           dx_c if s16 else dx, cdt, junk_g, junk_b, stream16=s16)
            del dpre, dh2
            op2 = dxm_c if dxm_c is not None else dxm
            # attention:  xm = x + attn(h1 Wqkv_eff^T + b) Wo^T + bo
            do = ops.gemm(op2, self._w(b + "self_attn.out_proj.weight", cdt), b_kn=True)
            dqkv = ops.attention_bwd(S["qkv"], S["o"], do, S["lse"], batch, seq, heads, d, d ** -0.5, L.MASK_NONE, None)
            if S["drop"] is None:
                # merged form: dW_eff = dY^T h1, dB = s dW_eff A^T, dA = s B^T dW_eff
                ops.gemm(dqkv, S["h1"], trans_a=True, b_kn=True, out=dw)
                for slot, n in enumerate("qkv"):
                    if f"{n}_proj" not in self.lora["targets"]:
                        continue
                    la, lb = self.lora_params(tower, i, f"{n}_proj")
                    dwp = dw[slot * w:(slot + 1) * w]
                    # 8-wide outputs: a handful of tiles, so the contraction is split to occupy more CUs
                    sk = 8 if w % 256 == 0 else 1
                    ops.gemm(dwp, la.data, alpha=scale, out=grad_view(lb), split_k=sk)                            # dW A^T
                    ops.gemm(lb.data, dwp, trans_a=True, b_kn=True, alpha=scale, out=grad_view(la), split_k=sk)  # B^T dW
                dh1 = ops.gemm(dqkv, self._packed[b + "self_attn.w"], b_kn=True)
            else:
                # unmerged form, per adapted projection with its own mask D (regenerated from the seed):
                #   y += s u B^T, u = (D h1) A^T   =>   du = s dy B,  dB = s dy^T u,  dA = du^T (D h1),
                #   dh1 += D (du A)
                dh1 = ops.gemm(dqkv, self._packed[b + "self_attn.w"], b_kn=True)      # base rows
                p_drop = self.lora["dropout"]
                pk = self._lora_packed.get(tower) if "u" in S["drop"] else None
                if pk is not None:
                    # csrc/lora.hip: every adapted projection of the layer per launch, each wide tensor read once per pass
                    u, sd = S["drop"]["u"], S["drop"]["seeds"]
                    r, offs, zero = pk["r"], pk["offs"], [0] * len(sd)
                    keep_inv = 1.0 / (1.0 - p_drop)
                    grads = [self.lora_params(tower, i, f"{n}_proj") for n in pk["names"]]
                    du = ops.lora_down(dqkv, offs, pk["bt"][i], r, w)                                # scaling dy B
                    ops.lora_tn(dqkv, offs, u, [grad_view(lb) for _, lb in grads], r, w, True, alpha=scale)
                    d_a = [grad_view(la) for la, _ in grads]
                    if len(sd) * r <= 16:      # dA = du^T (D . h1) and dh1 += D . (du A): one walk, one mask evaluation
                        ops.lora_bwd_input(S["h1"], du, pk["a"][i], dh1, d_a, r, w, p_drop, sd, alpha=keep_inv)
                    else:
                        ops.lora_tn(S["h1"], zero, du, d_a, r, w, False, p_drop, sd, alpha=keep_inv)
                        ops.lora_up(du, pk["a"][i], dh1, zero, r, w, p_drop, sd, alpha=keep_inv)
                for slot, n in enumerate("qkv"):
                    if pk is not None or n not in S["drop"]:
                        continue
                    u, seed = S["drop"][n]
                    la, lb = self.lora_params(tower, i, f"{n}_proj")
                    dy = dqkv[:, slot * w:(slot + 1) * w]
                    du = ops.gemm(dy, lb.data.to(cdt), b_kn=True, alpha=scale)                    # [M, r]
                    ops.gemm(dy, u, trans_a=True, b_kn=True, alpha=scale, out=grad_view(lb))      # [w, r]
                    hd = ops.dropout(S["h1"], p_drop, seed)
                    ops.gemm(du, hd, trans_a=True, b_kn=True, out=grad_view(la))                  # [r, w]
                    t = ops.gemm(du, la.data.to(cdt), b_kn=True)                                  # [M, w]
                    ops.dropout(t, p_drop, seed, out=dh1, accumulate=True)
                    del hd, t, du
            dx, dx_c = ops.layernorm_bwd(dh1, S["x"], self._f(b + "layer_norm1.weight"), S["m1"], S["r1"],
                                         dxm_c if s16 else dxm, cdt, junk_g, junk_b, stream16=s16)
            saved[i] = None                                # free this block's activations early
        # the embeddings below are frozen: the walk ends here

    def _final_norm(self, x, name: str, keep: bool):
        y, m, r = ops.layernorm_fwd(x, self._f(name + ".weight"), self._f(name + ".bias"), torch.float32,
                                    save_stats=keep, eps=LN_EPS)
        return y, (x, m, r)

    def _keep(self) -> bool:
        keep = self.lora is not None and torch.is_grad_enabled()
        self._unmerged_now = bool(keep and self.lora["dropout"] > 0.0 and self.training)
        return keep

    def vision_tokens(self, img: torch.Tensor, cdt: torch.dtype) -> torch.Tensor:
        """[N,3,S,S] fp32 -> post_layernorm(tokens) fp32 [N*(S/p)^2, W]."""
        keep = self._keep()
        with torch.no_grad():
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
            del cols
            layers: Optional[List] = [] if keep else None
            if self._frozen_fast(keep, self.vcfg, cdt):
                return self._encoder_frozen(x, "vision_model", self.vcfg, nimg, n, cdt, "vision_model.post_layernorm")
            x = self._encoder(x, "vision_model", self.vcfg, nimg, n, cdt, layers)
            y, fin = self._final_norm(x, "vision_model.post_layernorm", keep)
            if keep:
                self._saved["vision_model"] = dict(layers=layers, fin=fin, batch=nimg, seq=n, cdt=cdt)
        return y

    def text_tokens(self, ids: torch.Tensor, cdt: torch.dtype) -> torch.Tensor:
        """[B,L] int -> final_layer_norm(tokens) fp32 [B*L, W]."""
        keep = self._keep()
        with torch.no_grad():
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
            layers: Optional[List] = [] if keep else None
            if self._frozen_fast(keep, self.tcfg, cdt):
                return self._encoder_frozen(x, "text_model", self.tcfg, batch, ctx, cdt, "text_model.final_layer_norm")
            x = self._encoder(x, "text_model", self.tcfg, batch, ctx, cdt, layers)
            y, fin = self._final_norm(x, "text_model.final_layer_norm", keep)
            if keep:
                self._saved["text_model"] = dict(layers=layers, fin=fin, batch=batch, seq=ctx, cdt=cdt)
        return y

    @torch.no_grad()
    def backward(self, tower: str, d_tokens: torch.Tensor, grad_view) -> None:
        """Adapter gradients of one tower from d loss / d (its output tokens), fp32 [M, W] in the row
        order of the last vision_tokens / text_tokens call.  `grad_view(param)` names the buffer
        each gradient is WRITTEN to (the model's gradient arena)."""
        if not self.lora:
            raise RuntimeError("SiglipTowers.backward: the towers are frozen (lora=None)")
        S = self._saved.pop(tower, None)
        if S is None:
            raise RuntimeError(f"SiglipTowers.backward({tower}): no forward with gradients enabled preceded it")
        cfg = self.vcfg if tower == "vision_model" else self.tcfg
        fin_name = "vision_model.post_layernorm" if tower == "vision_model" else "text_model.final_layer_norm"
        cdt = S["cdt"]
        x, m, r = S["fin"]
        w = cfg["hidden_size"]
        junk_g = torch.empty(w, dtype=torch.float32, device=x.device)
        junk_b = torch.empty(w, dtype=torch.float32, device=x.device)
        dx, dx_c = ops.layernorm_bwd(d_tokens.contiguous(), x, self._f(fin_name + ".weight"), m, r, None, cdt,
                                     junk_g, junk_b)
        self._encoder_bwd(dx, dx_c, S["layers"], tower, cfg, S["batch"], S["seq"], cdt, grad_view)

    def check_token_ids(self) -> None:
        if self._err_flag is not None and int(self._err_flag.item()) != 0:
            self._err_flag.zero_()
            raise IndexError("instruction contains a token id outside the SigLIP vocabulary")


def from_pretrained(name: str, checkpoint: Optional[str] = None, lora: Optional[Dict] = None) -> SiglipTowers:
    """Stand-in for AutoModel.from_pretrained (no network): build the towers from the named
    checkpoint's hyper-parameters and fill them from a local copy when one is given: the
    Hugging Face snapshot directory, its model.safetensors, or a state dict with the same
    (transformers SiglipModel) key names -- see checkpoint_io.read_state_dict."""
    if name not in SIGLIP_CONFIGS:
        raise ValueError(f"unknown SigLIP model {name!r}; known: {sorted(SIGLIP_CONFIGS)}")
    cfg = SIGLIP_CONFIGS[name]
    model = SiglipTowers(cfg["vision"], cfg["text"], lora=lora)
    if checkpoint:
        from .checkpoint_io import load_into, read_state_dict
        # a published (HF-named) file fills the base weights of a LoRA-wrapped tower too; the adapters
        # keep their initialisation unless the file (a PEFT-named fine-tuned checkpoint) carries them
        sd = {model._real.get(k, k): v for k, v in read_state_dict(checkpoint).items()}
        load_into(model, sd, f"SigLIP {name}", optional_containing=(".lora_",))
    else:
        warnings.warn(f"SigLIP {name}: no checkpoint given, using random initialisation "
                      f"(set model.siglip_checkpoint=<snapshot dir | model.safetensors | state_dict.pt>)")
    return model.eval()

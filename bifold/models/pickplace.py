"""Fusion transformer + pick/place heatmap heads on libbifold_hip, with a hand-written
backward pass.

Parameter containers keep the reference's state-dict keys:
  pick_place.fusion.token_type_embeddings.weight                       (models/fusion.py:15)
  pick_place.fusion.transformer_encoder.layers.{i}.0.norm.{weight,bias}   PreNorm(Attention)
  ...layers.{i}.0.fn.to_qkv.weight / .0.fn.to_out.0.{weight,bias}     (models/transformer.py:19-24)
  ...layers.{i}.1.norm.* / .1.fn.net.0.* / .1.fn.net.3.*              (models/modules.py:21-27)
  pick_place.{pick,place | left_pick,...}_decoder.decoder_net.{0,2,4,6,8}.{weight,bias}
The compute is one autograd.Function for the whole trainable tail of RGBOnly:
project -> token assembly -> depth x (LN, QKV GEMM, attention, out-proj(+res), LN, FC1(+GELU),
FC2(+res)) -> collapsed decoder -> sigmoid; its backward replays the chain with fused
epilogues and writes parameter gradients straight into the flat gradient arena.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from bifold_amd.hip import lib as L
from bifold_amd.hip import ops
from . import Components
from .arena import ParamArena
from .gradscale import scaler_of
from .streams import serial, side_stream
from .decoder_math import upsample_operator
from .params import add_param, layernorm_init, linear_init

DEC_IDX = (0, 2, 4, 6, 8)
# Module switches below are CONSTANTS since round 6 (they were BIFOLD_* environment variables while their A/Bs ran; the
# numbers are in DESIGN.md section 6 and docs/).  Tests flip two of them in-process to hold both forms against each other.
# The fusion MLP saves gelu'(pre) for the backward (one erf evaluation in the forward epilogue, a multiply in the
# backward's); False keeps `pre` and evaluates gelu' in the backward (the round-1 form)
_SAVEGRAD = True
# the weight-gradient GEMMs (dW = dY^T X) have no consumer before the optimiser / the gradient all-reduce: they run on a
# side stream next to the dX chain, so that one persistent GEMM's under-filled last round is filled by the other's
# workgroups: the main stream joins the side stream at the end of every layer.  Under streams.serial() (the profiling mode
# of bench.py's roofline sample and tools/serial_stats.sh) everything runs on one stream.
# 1 = a dW GEMM is issued AFTER the dX GEMM that becomes runnable at the same moment (see fusion_layer_bwd); 0 = before
_DW_ORDER = 1
# 1 = the four dW GEMMs of a layer are collected and issued as ONE grouped launch of the persistent kernel once the last
# of their operands (dQKV) exists (bf_gemm_tn_grouped: 48 tiles x 5 K-splits for the 8 x 512 layer instead of four GEMMs
# split 16 / 16 / 64 / 21 ways, a quarter of the fp32 partial slabs, 2 launches instead of 8); groups that do not
# qualify (fp32 mode, short or ragged token counts) and 0 run the GEMMs one by one as before
_DW_GROUPED = 1


def _dw_stream(device) -> "torch.cuda.Stream":
    return side_stream(device, "dw")


class _WeightGradSide:
    """Launches `ops.gemm(dY, X, trans_a=True, b_kn=True, out=dW)` on the side stream once the main stream has produced
    its operands.  The operands were allocated on the main stream; the side stream must have read them before the
    caching allocator hands their memory out again.  Rounds 1-3 pinned them with `record_stream`; that makes the
    allocator re-check an event at every later allocation and take FRESH memory whenever the side stream is not done yet
    -- at configs[2]'s sizes (2.3 GB operands, ~9 GB per layer) the reserved pool grew to the whole 288 GB while 150-180
    GB were allocated, and a step then ran into allocator retries (112-196 samples/s from run to run).  Round 4: the
    object simply HOLDS the operands until the main stream has waited for the event behind their GEMMs (`mark` /
    `joined`): they are then freed on the stream that allocated them, behind the wait, and reusable at once."""

    def __init__(self, device):
        self.on = not serial()
        self.side = _dw_stream(device) if self.on else None
        self.group = [] if _DW_GROUPED else None       # (dy, x, out) of the layer being walked
        self.group_after = None                         # event behind the producer of the LAST collected operand
        self.qualifies: Optional[bool] = None           # does the grouped launch take this backward's token count?
        self.held: List[torch.Tensor] = []              # operands of side-stream GEMMs not yet behind a `mark`
        self.marked: List = []                          # (event, operands) generations waiting for `joined`

    def flush(self):
        """Issue what `gemm` collected (end of a layer: every operand has been produced on the main stream)."""
        if not self.group:
            return
        todo, self.group = self.group, []
        after, self.group_after = self.group_after, None
        if self.on:
            if after is not None:
                self.side.wait_event(after)             # not for what the main stream queued since (the next dX GEMM)
            else:
                self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side) if self.on else _NullCtx():
            if not ops.gemm_tn_grouped(todo):
                for dy, x, out in todo:
                    ops.gemm(dy, x, trans_a=True, b_kn=True, out=out)
        if self.on:
            for dy, x, _ in todo:
                self.held += [dy, x]

    def prepare(self, rows: int, dtype) -> None:
        """Decide once per backward whether the grouped launch takes this token count (every layer has the same)."""
        if self.group is not None and self.qualifies is None:
            self.qualifies = ops.gemm_tn_grouped_qualifies(rows, dtype)
            if not self.qualifies:
                self.group = None

    def ready(self, last: bool = False):
        """Event on the main stream: everything queued so far (the operands of the next dW GEMM) is produced.
        A layer whose dW GEMMs leave as ONE grouped launch needs only the event behind the LAST operand (`last`): an
        event record between two dependent kernels is a 7 us bubble on the main stream (profiles/r06_step_gaps.txt:
        three of the four records of a layer were never waited for), so the others are not recorded.  (The device's idle
        time in the backward fell from 255 to 117 us per step and the step did not move -- 7 926 vs 7 925 samples/s over
        four interleaved pairs: the part is at its power cap, an idle gap is not lost time.)"""
        if not self.on:
            return None
        if self.group is not None and self.qualifies and not last:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def gemm(self, dy, x, out, after=None):
        # decided once per backward (every layer has the same token count): a layer whose weight gradients the
        # grouped launch would refuse (few tokens, an odd count) keeps the per-GEMM issue order and lifetimes
        self.prepare(dy.shape[0], dy.dtype)
        if self.group is not None and dy.dtype in ops.HALF:
            self.group.append((dy, x, out))
            if after is not None:
                self.group_after = after
            return
        if not self.on:
            ops.gemm(dy, x, trans_a=True, b_kn=True, out=out)
            return
        if after is not None:
            self.side.wait_event(after)
        else:
            self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            ops.gemm(dy, x, trans_a=True, b_kn=True, out=out)
        self.held += [dy, x]

    def mark(self):
        """Event after which everything launched so far on the side stream is complete."""
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side)
        self.marked.append((ev, self.held))
        self.held = []
        return ev

    def joined(self, ev) -> None:
        """The main stream has been told to wait for `ev`: the operands of everything marked up to it may go."""
        while self.marked:
            e, _ = self.marked.pop(0)
            if e is ev:
                break


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def head_names(is_bimanual: bool) -> List[str]:
    """Decoder order of PickPlaceConvDecoder (models/pickplace.py:34-47)."""
    return ["left_pick", "right_pick", "left_place", "right_place"] if is_bimanual else ["pick", "place"]


class ConcatTransformer(nn.Module):
    """Parameter container of the concat-fusion transformer (models/fusion.py:8-33)."""

    def __init__(self, dim, heads, depth, dropout, mlp_ratio, num_modalities=2, num_registers=0, **kwargs):
        super().__init__()
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        # nn.Dropout(dropout) sits in four places of every block (ref:bifold/models/transformer.py:18-24,36,
        # ref:bifold/models/modules.py:21-27): on the attention probabilities, behind to_out, behind the GELU and behind the
        # MLP's second Linear.  Active in training only; masks are functions of (seed, element index) (bf_dropout), the
        # backward regenerates them, `last_dropout_seeds` lets a checker replay them.  All shipped configs say 0.0.
        self.dropout = float(dropout)
        self.last_dropout_seeds: Dict = {}
        if num_registers:
            raise NotImplementedError("register tokens are not used by any shipped config")
        if dim % heads:
            raise ValueError("dim must be divisible by heads")
        self.dim, self.heads, self.depth = dim, heads, depth
        self.compute_dtype = torch.bfloat16       # standalone forward only; the models pass their own dtype
        self.dim_head = dim // heads
        self.mlp_dim = dim * mlp_ratio
        self.num_modalities = num_modalities
        add_param(self, "token_type_embeddings.weight", torch.randn(num_modalities, dim) * 0.02)
        for i in range(depth):
            b = f"transformer_encoder.layers.{i}."
            for n in ("0.norm", "1.norm"):
                for k, v in layernorm_init(dim).items():
                    add_param(self, f"{b}{n}.{k}", v)
            add_param(self, b + "0.fn.to_qkv.weight", linear_init(3 * dim, dim, bias=False)["weight"])
            for k, v in linear_init(dim, dim).items():
                add_param(self, f"{b}0.fn.to_out.0.{k}", v)
            for k, v in linear_init(self.mlp_dim, dim).items():
                add_param(self, f"{b}1.fn.net.0.{k}", v)
            for k, v in linear_init(dim, self.mlp_dim).items():
                add_param(self, f"{b}1.fn.net.3.{k}", v)

    def layer_params(self, i: int) -> Dict[str, nn.Parameter]:
        lay = self.transformer_encoder.layers._modules[str(i)]
        a, f = lay._modules["0"], lay._modules["1"]
        return dict(g1=a.norm.weight, b1=a.norm.bias, wqkv=a.fn.to_qkv.weight,
                    wo=a.fn.to_out._modules["0"].weight, bo=a.fn.to_out._modules["0"].bias,
                    g2=f.norm.weight, b2=f.norm.bias, w1=f.fn.net._modules["0"].weight,
                    bb1=f.fn.net._modules["0"].bias, w2=f.fn.net._modules["3"].weight,
                    bb2=f.fn.net._modules["3"].bias)

    def forward(self, *inputs, modalities=None, attention_masks=None, return_last_input: bool = False, **kwargs):
        """Standalone use of the component (models/fusion.py:35-64): fp32 CUDA inputs [B, n_i, dim] ->
        (features of the LAST input's tokens [B, n_last, dim], None).  Like the reference the type
        embedding is added to the inputs IN PLACE (fusion.py:52).  The models never call this: RGBOnly /
        SigLip run the same layers inside their fused tail (RGBTailFn / SegmentTailFn), which also fuses
        the token assembly and the decoders.

        `return_last_input` (not a reference argument; PickPlaceConvDecoder's mask head uses it): the second value is the
        last input as the layers saw it -- type embedding added -- as a tensor of THIS autograd node, so that a gradient
        into it reaches the input and the type embedding like the reference's in-place add makes it."""
        if modalities is None:
            modalities = range(len(inputs))
        modalities = [int(m) for m in modalities]
        assert len(inputs) == len(modalities)
        if any(not t.is_cuda for t in inputs):
            raise RuntimeError("ConcatTransformer runs on the MI355X only (there is no CPU path)")
        mask = None
        if attention_masks is not None:
            mask = attention_masks.to(torch.int32).contiguous()
        with torch.no_grad():                              # the reference's in-place `inp += type_embeddings`;
            for mod, inp in zip(modalities, inputs):       # its gradient is produced by _FusionStackFn.backward
                inp.add_(self.token_type_embeddings.weight[mod].to(inp.dtype))
        out, last_in = _FusionStackFn.apply(self, tuple(modalities), mask, len(inputs), *inputs, *self.parameters())
        return out[:, -inputs[-1].shape[1]:, :], (last_in if return_last_input else None)


class PickPlaceConvDecoder(nn.Module):
    def __init__(self, dim, is_bimanual, fusion_model, num_patches, patch_size, compute_mask=False,
                 detach_mask=False, **kwargs):
        super().__init__()
        # compute_mask (ref:bifold/models/pickplace.py:25-30): one more ConvDecoder, on the image tokens the fusion model is
        # HANDED; its sigmoid is `mask_heatmap` and multiplies every pick heatmap (detach_mask: as a constant).  Off in
        # every shipped model YAML; conf/loss/bce_mask.yaml / composed*.yaml are the losses that read it.
        self.has_mask = bool(compute_mask)
        self.detach_mask = bool(detach_mask)
        self.is_bimanual = is_bimanual
        self.dim = dim
        self.num_patches = num_patches
        self.grid = int(math.isqrt(num_patches))
        if self.grid * self.grid != num_patches:
            raise ValueError("num_patches must be a perfect square")
        self.fusion = Components.get_by_name(fusion_model, dim=dim, **kwargs)
        self.heads_out = head_names(is_bimanual)
        self.n_pick = len(self.heads_out) // 2              # the pick heads lead (head_names)
        # every collapsed decoder of the component, in buffer order: the heads, then the mask head
        self.decoders_all = self.heads_out + (["mask"] if self.has_mask else [])
        chans = [dim, dim // 2, dim // 2, dim // 4, dim // 4, 1]
        # registration order = the reference's (fusion, mask_head, the heads): checkpoints index Adam's state by it
        for name in (["mask"] if self.has_mask else []) + self.heads_out:
            mod = self._decoder_module(name)
            for j, i in enumerate(DEC_IDX):
                init = linear_init(chans[j + 1], chans[j])
                add_param(self, f"{mod}.decoder_net.{i}.weight", init["weight"].view(chans[j + 1], chans[j], 1, 1))
                add_param(self, f"{mod}.decoder_net.{i}.bias", init["bias"])
        self._U: Optional[torch.Tensor] = None
        self._collapsed = None

    @staticmethod
    def _decoder_module(name: str) -> str:
        return "mask_head" if name == "mask" else f"{name}_decoder"

    def decoder_params(self, name: str):
        net = self._modules[self._decoder_module(name)].decoder_net
        ws = [net._modules[str(i)].weight for i in DEC_IDX]
        bs = [net._modules[str(i)].bias for i in DEC_IDX]
        return ws, bs

    # The collapse of a head's five 1x1 convs into (w_eff, b_eff) and its backward are single-workgroup kernels
    # (41 / 180 us each, two or four heads): latency, not work.  They run on a side HIP stream -- the forward one
    # at the top of the model's forward, beside the towers; the backward one beside the fusion layers' backward.
    def _side(self, device) -> "torch.cuda.Stream":
        return side_stream(device, "decoder")

    def collapse_async(self) -> None:
        """Start decoder_collapse_fwd for every head on the side stream (after everything already queued on the
        current stream, i.e. after the optimiser step that produced the weights); decoder_heads_fwd joins it."""
        ws0, _ = self.decoder_params(self.heads_out[0])
        dev = ws0[0].device
        main, side = torch.cuda.current_stream(), self._side(dev)
        dim, nh = self.dim, len(self.decoders_all)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            w_eff = torch.empty(nh, dim, dtype=torch.float32, device=dev)
            b_eff = torch.empty(nh, dtype=torch.float32, device=dev)
            rows_ws = torch.empty(nh, 4 * dim, dtype=torch.float32, device=dev)
            for h, name in enumerate(self.decoders_all):
                ws, bs = self.decoder_params(name)
                ops.decoder_collapse_fwd([_w2d(w.data) for w in ws], [b.data for b in bs], w_eff[h], b_eff[h:h + 1],
                                         rows_ws[h])
            ev = torch.cuda.Event()
            ev.record(side)
        self._collapsed = (w_eff, b_eff, rows_ws, ev)

    def upsample_matrix(self, device) -> torch.Tensor:
        if self._U is None or self._U.device != device:
            self._U = upsample_operator(self.grid).to(device)
        return self._U

    def forward(self, *inputs, **kwargs):
        """Standalone use of the component (models/pickplace.py:59-124): `*inputs` / `**kwargs` go to the
        fusion model, the last input's tokens (minus its lead token) feed the decoders.  Returns the
        reference's dict: "attn_weights" (None) and one `[B, S, S]` heatmap per head."""
        component_arena(self)                              # one arena for fusion + decoders, before either runs
        dec_params = [p for n, p in self.named_parameters() if "_decoder." in n or "mask_head." in n]
        if self.has_mask:
            # the mask head reads the last input as the fusion model leaves it: type embedding added in place
            # (models/fusion.py:52, models/pickplace.py:66-69)
            fused, last_in = self.fusion(*inputs, return_last_input=True, **kwargs)
            heats = _DecoderHeadsFn.apply(self, fused.contiguous(), last_in, *dec_params)
        else:
            fused, _ = self.fusion(*inputs, **kwargs)
            heats = _DecoderHeadsFn.apply(self, fused.contiguous(), None, *dec_params)
        out = {"attn_weights": None}
        for name, h in zip(self.heads_out, heats):
            out[f"{name}_heatmap"] = h
        if self.has_mask:
            out["mask_heatmap"] = heats[len(self.heads_out)]
        return out


# ======================================================================================
#                                 fused tail: forward
# ======================================================================================
class _Ctx:
    """What the backward needs; plain attributes instead of save_for_backward because the
    buffers are never exposed to autograd (they are produced and consumed by HIP kernels)."""
    pass


def _w2d(t: torch.Tensor) -> torch.Tensor:
    return t.view(t.shape[0], -1)


# 1 (16-bit modes): the residual branches (out-proj, fc2) leave the GEMM as 16-bit tensors and the residual add rides in
# the NEXT LayerNorm's pass (bf_add_layernorm_fwd, what the towers do) instead of in the GEMM's fp32 epilogue: the
# fp32-residual epilogue costs a 512-wide GEMM 57 - 61 us per launch (a 256 KB fp32 tile read and written per item), the
# add inside the LayerNorm pass 40 (tools/decoder_bench.py's neighbour).  0: round 1-4's form.
_ADD_IN_LN = 1
# 1 (fp16 mode): the gradient of the residual stream travels through the fusion stack's backward as ONE fp16 tensor --
# it is what the dX GEMMs read anyway -- instead of an fp32 tensor plus its fp16 copy: the two LayerNorm backward passes
# of a block read 72 instead of 144 MB of it and write 72 instead of 216 (bf_layernorm_bwd without an fp32 dx).  Every
# add rounds to 11 bits of the running gradient (the loss scale keeps it in range like every other fp16 gradient of the
# step); bf16's 8 bits would show in the gradient gates, so bf16 and fp32 keep the fp32 stream.  0: round 1-4's form.
_GRAD_STREAM16 = 1


def fusion_layer_fwd(x, P, A: ParamArena, cdt, batch, seq, heads, mask_mode, row_mask, save: bool, pending=None,
                     last: bool = True, drop=None):
    """One pre-LN block on the fp32 residual stream x [B*seq, dim].  `pending` (16-bit, optional) is the previous block's
    MLP branch, not yet added to x: it is added in this block's first LayerNorm pass.  Returns (x_out, saved, pending_out):
    with `last` False (and a 16-bit compute dtype) the block's own MLP branch comes back as `pending_out` instead of being
    added to x_out."""
    d = x.shape[1] // heads
    defer = bool(_ADD_IN_LN) and cdt in ops.HALF and x.shape[1] in (256, 512, 768, 1024)   # bf_add_layernorm_fwd's widths
    if pending is None:
        h1, m1, r1 = ops.layernorm_fwd(x, P["g1"].data, P["b1"].data, cdt, save_stats=save)
    else:                                    # x <- x + pending, h1 = LN1(x): one pass; a new x (the old one is saved)
        h1, m1, r1, x = ops.add_layernorm_fwd(x, pending, P["g1"].data, P["b1"].data, cdt, x_out=torch.empty_like(x),
                                              save_stats=save)
    qkv = ops.gemm(h1, A.operand(P["wqkv"], cdt))
    # `drop` = (p, seeds) in a training forward with dropout > 0: the four nn.Dropout sites of the block
    pd, sd = drop if drop is not None else (0.0, {})
    o, lse = ops.attention_fwd(qkv, batch, seq, heads, d, d ** -0.5, mask_mode, row_mask, need_lse=save,
                               dropout=pd, seed=sd.get("attn", 0))
    if defer:
        branch = ops.gemm(o, A.operand(P["wo"], cdt), bias=P["bo"].data)
        if drop is not None:
            ops.dropout(branch, pd, sd["out"], out=branch)
        h2, m2, r2, x2 = ops.add_layernorm_fwd(x, branch, P["g2"].data, P["b2"].data, cdt, x_out=torch.empty_like(x),
                                               save_stats=save)
        del branch
    else:
        if drop is not None:                       # x2 = x + dropout(o Wo^T + bo): the branch in fp32, masked into a copy of x
            branch = ops.gemm(o, A.operand(P["wo"], cdt), bias=P["bo"].data, out_dtype=torch.float32)
            x2 = x.clone()
            ops.dropout(branch, pd, sd["out"], out=x2, accumulate=True)
            del branch
        else:
            x2 = ops.gemm(o, A.operand(P["wo"], cdt), bias=P["bo"].data, residual=x, out_dtype=torch.float32)
        h2, m2, r2 = ops.layernorm_fwd(x2, P["g2"].data, P["b2"].data, cdt, save_stats=save)
    if save:       # the forward saves gelu'(pre), not pre: one erf evaluation serves both outputs of the epilogue,
        # and the backward GEMM's epilogue is a multiply instead of an erf + exp per element
        a, a_pre = ops.gemm(h2, A.operand(P["w1"], cdt), bias=P["bb1"].data,
                            act=L.ACT_ERF_GELU_SAVEGRAD if _SAVEGRAD else L.ACT_ERF_GELU, want_pre=True)
    else:
        a, a_pre = ops.gemm(h2, A.operand(P["w1"], cdt), bias=P["bb1"].data, act=L.ACT_ERF_GELU), None
    if drop is not None:
        # behind the GELU: the SAME mask goes onto the activation and onto the saved gelu' (the backward's multiplier then
        # carries D / (1 - p) with it)
        ops.dropout(a, pd, sd["act"], out=a)
        if a_pre is not None:
            if not _SAVEGRAD:
                raise RuntimeError("dropout > 0 needs the saved-gelu' form of the MLP backward")
            ops.dropout(a_pre, pd, sd["act"], out=a_pre)
    pending_out = None
    if defer and not last:
        x3, pending_out = x2, ops.gemm(a, A.operand(P["w2"], cdt), bias=P["bb2"].data)
        if drop is not None:
            ops.dropout(pending_out, pd, sd["ff"], out=pending_out)
    elif drop is not None:                         # x3 = x2 + dropout(a W2^T + b2)
        branch = ops.gemm(a, A.operand(P["w2"], cdt), bias=P["bb2"].data, out_dtype=torch.float32)
        x3 = x2.clone()
        ops.dropout(branch, pd, sd["ff"], out=x3, accumulate=True)
        del branch
    else:
        x3 = ops.gemm(a, A.operand(P["w2"], cdt), bias=P["bb2"].data, residual=x2, out_dtype=torch.float32)
    saved = None
    if save:
        saved = dict(x=x, m1=m1, r1=r1, h1=h1, qkv=qkv, o=o, lse=lse, x2=x2, m2=m2, r2=r2, h2=h2,
                     a_pre=a_pre, a=a, drop=drop)
    return x3, saved, pending_out


def fusion_layer_bwd(dx3, dx3_c, S, P, A: ParamArena, cdt, batch, seq, heads, mask_mode, row_mask, wg=None):
    """dx3 fp32 [M, D] (+ its cdt copy or None in fp32 mode) -> (dx, dx_c) for the layer input; dx3 None = the 16-bit
    gradient stream (_GRAD_STREAM16): dx3_c alone carries the gradient and (None, dx_c) comes back.
    `wg`: the _WeightGradSide that launches the dW GEMMs (default: in line on the current stream)."""
    d = (dx3_c if dx3 is None else dx3).shape[1] // heads
    if wg is None:
        wg = _WeightGradSide.__new__(_WeightGradSide)
        wg.on, wg.side, wg.group, wg.group_after, wg.qualifies, wg.held, wg.marked = False, None, None, None, None, [], []
    g = A.grad_view
    op3 = dx3_c if dx3_c is not None else dx3
    wg.prepare(op3.shape[0], op3.dtype)
    drop = S.get("drop")
    pd, sd = drop if drop is not None else (0.0, {})
    if drop is not None:
        op3 = ops.dropout(op3, pd, sd["ff"])       # through dropout(a W2^T + b2): the mask of the forward, regenerated
    # FF:  x3 = x2 + gelu(h2 W1^T + b1) W2^T + b2
    # Every dW GEMM waits only for the producer of its operands (an event recorded right behind it) but is ISSUED after the
    # next dX GEMM of the chain (_DW_ORDER = 1): both become runnable at the same moment, and the one the hardware sees
    # first takes the CUs -- it should be the one the rest of the backward waits for.  (_DW_ORDER = 0: dW first.)
    e3 = wg.ready()
    if not _DW_ORDER:
        wg.gemm(op3, S["a"], g(P["w2"]), e3)
    da_pre = ops.gemm(op3, A.operand(P["w2"], cdt), b_kn=True,
                      act=L.ACT_MUL_AUX if _SAVEGRAD else L.ACT_ERF_GELU_GRAD, aux=S["a_pre"],
                      colsum_out=g(P["bb1"]))          # d bias1 = column sums of da_pre, fused into the epilogue
    if _DW_ORDER:
        wg.gemm(op3, S["a"], g(P["w2"]), e3)
    e_pre = wg.ready()
    if not _DW_ORDER:
        wg.gemm(da_pre, S["h2"], g(P["w1"]), e_pre)
    dh2 = ops.gemm(da_pre, A.operand(P["w1"], cdt), b_kn=True)
    if _DW_ORDER:
        wg.gemm(da_pre, S["h2"], g(P["w1"]), e_pre)
    # the LayerNorm backward streams dx3 (in) and dx2 (out): it also emits their column sums,
    # i.e. the bias gradients of net.3 and to_out
    s16 = dx3 is None                                       # the 16-bit gradient stream (_GRAD_STREAM16): dx3_c is all there is
    dx2, dx2_c = ops.layernorm_bwd(dh2, S["x2"], P["g2"].data, S["m2"], S["r2"], dx3_c if s16 else dx3, cdt,
                                   g(P["g2"]), g(P["b2"]), dres_sum_out=g(P["bb2"]), dx_sum_out=g(P["bo"]), stream16=s16)
    op2 = dx2_c if dx2_c is not None else dx2
    if drop is not None:
        # the two bias gradients the LayerNorm pass just summed are those of the UNMASKED gradients: redo them on the masked ones
        ops.colsum(op3, out=g(P["bb2"]))
        op2 = ops.dropout(op2, pd, sd["out"])      # through dropout(o Wo^T + bo)
        ops.colsum(op2, out=g(P["bo"]))
    # attention:  x2 = x + attn(h1 Wqkv^T) Wo^T + bo
    e2 = wg.ready()
    if not _DW_ORDER:
        wg.gemm(op2, S["o"], g(P["wo"]), e2)
    do = ops.gemm(op2, A.operand(P["wo"], cdt), b_kn=True)
    if _DW_ORDER:
        wg.gemm(op2, S["o"], g(P["wo"]), e2)
    dqkv = ops.attention_bwd(S["qkv"], S["o"], do, S["lse"], batch, seq, heads, d, d ** -0.5, mask_mode, row_mask,
                             dropout=pd, seed=sd.get("attn", 0))
    e_qkv = wg.ready(last=True)
    if not _DW_ORDER:
        wg.gemm(dqkv, S["h1"], g(P["wqkv"]), e_qkv)
    dh1 = ops.gemm(dqkv, A.operand(P["wqkv"], cdt), b_kn=True)
    if _DW_ORDER:
        wg.gemm(dqkv, S["h1"], g(P["wqkv"]), e_qkv)
    wg.flush()
    return ops.layernorm_bwd(dh1, S["x"], P["g1"].data, S["m1"], S["r1"], dx2_c if s16 else dx2, cdt, g(P["g1"]),
                             g(P["b1"]), stream16=s16)


def fusion_stack_fwd(fus: "ConcatTransformer", A: ParamArena, cdt, x, batch: int, seq: int, mask_mode: int,
                     row_mask, save: bool):
    """depth x fusion_layer_fwd on x fp32 [B*seq, dim]; returns (x_out, per-layer saved activations)."""
    layers = []
    pending = None
    p_drop = float(getattr(fus, "dropout", 0.0))
    active = p_drop > 0.0 and fus.training and save
    if active:
        # one host-side draw per forward from torch's CPU generator (seeded by cfg.seed): no device sync; the rank is
        # folded in so that data-parallel replicas drop different elements
        base = int(torch.empty((), dtype=torch.int64).random_().item())
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            base ^= (torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
        fus.last_dropout_seeds = {}
    for i in range(fus.depth):
        drop = None
        if active:
            seeds = {n: (base + 8 * i + j) & (2 ** 63 - 1) for j, n in enumerate(("attn", "out", "act", "ff"))}
            fus.last_dropout_seeds[i] = seeds
            drop = (p_drop, seeds)
        x, s, pending = fusion_layer_fwd(x, fus.layer_params(i), A, cdt, batch, seq, fus.heads, mask_mode, row_mask, save,
                                         pending=pending, last=i == fus.depth - 1, drop=drop)
        layers.append(s)
    return x, layers


def fusion_stack_bwd(fus: "ConcatTransformer", A: ParamArena, cdt, dx, layers, batch: int, seq: int, mask_mode: int,
                     row_mask, on_layer_done=None):
    """Backward of fusion_stack_fwd: dx fp32 [B*seq, dim] w.r.t. the stack's output -> w.r.t. its input;
    every layer's parameter gradients go straight into the arena."""
    s16 = bool(_GRAD_STREAM16) and cdt == torch.float16
    if dx.dtype == cdt and cdt != torch.float32:           # already in the compute dtype (fusion_decoder_bwd, fp16 mode)
        dx_c = dx
    else:
        dx_c = ops.cast(dx, cdt) if cdt != torch.float32 else None
    keep16 = s16 and dx.dtype == cdt                       # ... and the caller takes the result in it too
    if s16:
        dx = None
    wg = _WeightGradSide(dx_c.device if s16 else dx.device)
    main = torch.cuda.current_stream() if wg.on else None
    for i in reversed(range(fus.depth)):
        dx, dx_c = fusion_layer_bwd(dx, dx_c, layers[i], fus.layer_params(i), A, cdt, batch, seq, fus.heads,
                                    mask_mode, row_mask, wg)
        layers[i] = None                                   # free this layer's activations early
        done = [i]
        if wg.on:                                          # (joining one layer late measured no better: removed in round 6)
            ev = wg.mark()
            main.wait_event(ev)
            wg.joined(ev)
        if on_layer_done is not None:
            for j in done:
                on_layer_done(j)
    if s16:
        dx = dx_c if keep16 else ops.cast(dx_c, torch.float32)   # standalone callers hand in and take fp32
    return dx


def decoder_heads_fwd(pp: "PickPlaceConvDecoder", tok: torch.Tensor, tok_bstride: int, batch: int,
                      mask_tok: Optional[torch.Tensor] = None, mask_bstride: int = 0):
    """tok: fp32 view [B, n_patches+1, dim] of the image-modality tokens (batch stride tok_bstride
    elements, used in place) -> heat [n_heads, B, S, S] + (w_eff, rows_ws) for the backward
    (models/pickplace.py:71-123 with ConvDecoder collapsed, models/modules.py:58-71).

    With the mask head (pp.has_mask; models/pickplace.py:66-69,76-98,107-119) `mask_tok` is the same kind of view of the
    tokens the fusion model was HANDED (type embedding included) and a fourth value comes back: dict(mask=sigmoid of the
    mask head [B, S, S], pick=heat[:n_pick] * mask, what the pick heads return); `heat` stays the plain sigmoids."""
    dim, nh, nall = pp.dim, len(pp.heads_out), len(pp.decoders_all)
    dev = tok.device
    pending, pp._collapsed = pp._collapsed, None
    if pending is not None:                                # started by collapse_async on the side stream
        w_eff, b_eff, rows_ws, ev = pending
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        for t in (w_eff, b_eff, rows_ws):
            t.record_stream(main)
    else:
        w_eff = torch.empty(nall, dim, dtype=torch.float32, device=dev)
        b_eff = torch.empty(nall, dtype=torch.float32, device=dev)
        rows_ws = torch.empty(nall, 4 * dim, dtype=torch.float32, device=dev)
        for h, name in enumerate(pp.decoders_all):
            ws, bs = pp.decoder_params(name)
            ops.decoder_collapse_fwd([_w2d(w.data) for w in ws], [b.data for b in bs], w_eff[h], b_eff[h:h + 1],
                                     rows_ws[h])
    low = ops.decoder_project(tok, tok_bstride, w_eff[:nh], b_eff[:nh], batch, nh, pp.num_patches, dim)
    heat = ops.decoder_upsample_sigmoid(low, pp.upsample_matrix(dev), nh * batch, pp.grid)
    size = pp.grid * 16
    heat = heat.view(nh, batch, size, size)
    if not pp.has_mask:
        return heat, w_eff, rows_ws
    if mask_tok is None:
        raise RuntimeError("compute_mask=True: the mask head needs the tokens the fusion model was handed")
    low_m = ops.decoder_project(mask_tok, mask_bstride, w_eff[nh:], b_eff[nh:], batch, 1, pp.num_patches, dim)
    mask = ops.decoder_upsample_sigmoid(low_m, pp.upsample_matrix(dev), batch, pp.grid).view(batch, size, size)
    return heat, w_eff, rows_ws, dict(mask=mask, pick=ops.mask_mul_fwd(heat[:pp.n_pick], mask))


def decoder_outputs(pp: "PickPlaceConvDecoder", heat, extra):
    """The component's heatmaps in `pp.heads_out` order (+ `mask_heatmap` last with the mask head)."""
    if extra is None:
        return tuple(heat[h] for h in range(len(pp.heads_out)))
    return (tuple(extra["pick"][h] for h in range(pp.n_pick)) +
            tuple(heat[h] for h in range(pp.n_pick, len(pp.heads_out))) + (extra["mask"],))


def decoder_heads_bwd(pp: "PickPlaceConvDecoder", A: ParamArena, tok, tok_bstride: int, heat, w_eff, rows_ws,
                      d_heats, d_tok, dtok_bstride: int, batch: int, grad_scale=None, extra=None, mask_tok=None,
                      mask_bstride: int = 0):
    """Backward of decoder_heads_fwd: writes d_tok (fp32 view like tok; token 0 gets zeros) and every
    decoder parameter gradient into the arena.  Returns the event after which the decoder gradients are
    complete: the per-head collapse backward runs on the side stream.  `grad_scale` (fp16 mode, models/gradscale.py)
    multiplies the low-resolution logit gradients, i.e. everything this function and its callers produce.

    With the mask head (`extra` of the forward; d_heats then carries d mask_heatmap behind the heads' gradients) the
    return value is (event, d_mask_tok): the gradient w.r.t. `mask_tok`, fp32 [B, n_patches+1, dim] (token 0 zeros),
    or None when nothing reaches the mask head (detach_mask and a loss that does not read mask_heatmap)."""
    g = A.grad_view
    dim, nh, nall = pp.dim, len(pp.heads_out), len(pp.decoders_all)
    dev = tok.device
    U = pp.upsample_matrix(dev)
    d_mask = None
    if extra is not None:
        d_mask_in = d_heats[nh] if len(d_heats) > nh else None
        through = not pp.detach_mask and any(d is not None for d in d_heats[:pp.n_pick])
        d_s, d_mask = ops.mask_mul_bwd(list(d_heats[:pp.n_pick]), heat[:pp.n_pick], extra["mask"], d_mask_in,
                                       pp.detach_mask, want_d_mask=through or d_mask_in is not None)
        d_heats = [d_s[h] for h in range(pp.n_pick)] + list(d_heats[pp.n_pick:nh])
    g_low = torch.empty(nh, batch, pp.grid * pp.grid, dtype=torch.float32, device=dev)
    for h in range(nh):
        if d_heats[h] is None:                             # a head the loss ignores
            g_low[h].zero_()
        else:
            ops.decoder_upsample_sigmoid_bwd(heat[h], d_heats[h].contiguous(), U, batch, pp.grid, out=g_low[h])
    if grad_scale is not None:                             # a number or the scaler's device scalar
        ops.scale_check_(g_low, grad_scale)
    g_w = torch.empty(nall, dim, dtype=torch.float32, device=dev)
    g_b = torch.empty(nall, dtype=torch.float32, device=dev)
    ops.decoder_project_bwd(tok, tok_bstride, w_eff[:nh], g_low, d_tok, dtok_bstride, g_w[:nh], g_b[:nh], batch, nh,
                            pp.num_patches, dim)
    d_mask_tok = None
    if extra is not None:
        if d_mask is None:                                 # nothing reaches the mask head: its gradients are zeros
            g_w[nh:].zero_()
            g_b[nh:].zero_()
        else:
            g_low_m = ops.decoder_upsample_sigmoid_bwd(extra["mask"], d_mask, U, batch, pp.grid)
            if grad_scale is not None:
                ops.scale_check_(g_low_m, grad_scale)
            d_mask_tok = torch.empty(batch, pp.num_patches + 1, dim, dtype=torch.float32, device=dev)
            ops.decoder_project_bwd(mask_tok, mask_bstride, w_eff[nh:], g_low_m, d_mask_tok, (pp.num_patches + 1) * dim,
                                    g_w[nh:], g_b[nh:], batch, 1, pp.num_patches, dim)
    main, side = torch.cuda.current_stream(), pp._side(dev)
    side.wait_stream(main)                                 # g_w / g_b are ready
    with torch.cuda.stream(side):
        for h, name in enumerate(pp.decoders_all):
            ws, bs = pp.decoder_params(name)
            ops.decoder_collapse_bwd([_w2d(w.data) for w in ws], [b.data for b in bs], rows_ws[h], g_w[h],
                                     g_b[h:h + 1], [_w2d(g(w)) for w in ws], [g(b) for b in bs])
        done = torch.cuda.Event()
        done.record(side)
    for t in (g_w, g_b, rows_ws):
        t.record_stream(side)
    return done if extra is None else (done, d_mask_tok)


def fusion_decoder_fwd(pp: "PickPlaceConvDecoder", A: ParamArena, cdt, x, batch: int, seq: int,
                       mask_mode: int, row_mask, save: bool):
    """x fp32 [B*seq, dim] (the assembled fusion input) -> the component's heatmaps (decoder_outputs: one [B, S, S] per
    head, `mask_heatmap` last with the mask head) and what the backward needs.  The LAST n_patches+1 tokens of
    every sequence are the image modality whose tokens feed the decoders (models/fusion.py:62-63,
    models/pickplace.py:71); the mask head reads the same rows of x ITSELF (models/pickplace.py:66-69)."""
    dim = pp.dim
    x_in = x
    x, layers = fusion_stack_fwd(pp.fusion, A, cdt, x, batch, seq, mask_mode, row_mask, save)
    first = seq - (pp.num_patches + 1)
    img_tok = x.view(batch, seq, dim)[:, first:]           # used in place through its batch stride
    extra = None
    if pp.has_mask:
        heat, w_eff, rows_ws, extra = decoder_heads_fwd(pp, img_tok, seq * dim, batch,
                                                        x_in.view(batch, seq, dim)[:, first:], seq * dim)
    else:
        heat, w_eff, rows_ws = decoder_heads_fwd(pp, img_tok, seq * dim, batch)
    saved = None
    if save:
        saved = dict(layers=layers, x_out=x, w_eff=w_eff, rows_ws=rows_ws, heat=heat, batch=batch, seq=seq,
                     mask_mode=mask_mode, row_mask=row_mask, extra=extra, x_in=x_in if pp.has_mask else None,
                     detach_mask=pp.detach_mask)
    return decoder_outputs(pp, heat, extra), saved


def fusion_decoder_bwd(pp: "PickPlaceConvDecoder", A: ParamArena, cdt, S, d_heats, on_layer_done=None,
                       grad_scale=None):
    """Backward of fusion_decoder_fwd: writes every fusion / decoder parameter gradient into the
    arena and returns d(fusion input) fp32 [B*seq, dim] (all of it times `grad_scale`)."""
    dim = pp.dim
    batch, seq = S["batch"], S["seq"]
    # fp16 mode: the stack's backward carries this gradient in fp16 (_GRAD_STREAM16) -- it is born that way
    s16 = bool(_GRAD_STREAM16) and cdt == torch.float16
    dx = torch.zeros(batch * seq, dim, dtype=cdt if s16 else torch.float32, device=S["x_out"].device)
    first = seq - (pp.num_patches + 1)
    xv = S["x_out"].view(batch, seq, dim)
    dxv = dx.view(batch, seq, dim)
    d_mask_tok = None
    if S.get("extra") is not None:
        dec_done, d_mask_tok = decoder_heads_bwd(pp, A, xv[:, first:], seq * dim, S["heat"], S["w_eff"], S["rows_ws"],
                                                 d_heats, dxv[:, first:], seq * dim, batch, grad_scale, S["extra"],
                                                 S["x_in"].view(batch, seq, dim)[:, first:], seq * dim)
    else:
        dec_done = decoder_heads_bwd(pp, A, xv[:, first:], seq * dim, S["heat"], S["w_eff"], S["rows_ws"], d_heats,
                                     dxv[:, first:], seq * dim, batch, grad_scale)
    joined = []

    def layer_done(i):             # the decoders' gradients leave with the first bucket: join before it is reported
        if not joined:
            torch.cuda.current_stream().wait_event(dec_done)
            joined.append(True)
        if on_layer_done is not None:
            on_layer_done(i)
    dx = fusion_stack_bwd(pp.fusion, A, cdt, dx, S["layers"], batch, seq, S["mask_mode"], S["row_mask"], layer_done)
    if not joined:
        torch.cuda.current_stream().wait_event(dec_done)
    if d_mask_tok is not None:     # the mask head read the fusion INPUT's image rows: its token gradient joins dx there
        ops.rows_accumulate(d_mask_tok, dx.view(batch, seq, dim)[:, first:])
    return dx


def component_arena(module: nn.Module) -> ParamArena:
    """The arena behind a component used on its own: the model's arena when the parameters already live
    in one (a component taken out of a built model), otherwise a new arena over the module's parameters."""
    params = [p for p in module.parameters()]
    a = getattr(params[0], "_bf_arena", None)
    if a is not None and a.grad is not None and a.still_bound() and all(getattr(p, "_bf_arena", None) is a
                                                                        for p in params):
        return a
    return ParamArena(list(module.named_parameters()), with_grad=True, with_shadow=True)


class _FusionStackFn(torch.autograd.Function):
    """ConcatTransformer.forward on its own: type-embedding add + concat (torch glue, a few MB) around the
    HIP fusion layers.  Gradients of the inputs are returned to autograd; parameter gradients are
    written into the arena (overwritten per backward, like the fused tails)."""

    @staticmethod
    def forward(ctx, fus, modalities, row_mask, n_in, *tensors):
        inputs = tensors[:n_in]
        A = component_arena(fus)
        A.sync_shadow()
        cdt = fus.compute_dtype
        x = torch.cat([t.float() for t in inputs], dim=1)
        batch, seq, dim = x.shape
        save = any(ctx.needs_input_grad)
        mode = L.MASK_QUERY_ROW if row_mask is not None else L.MASK_NONE
        y, layers = fusion_stack_fwd(fus, A, cdt, x.view(batch * seq, dim), batch, seq, mode, row_mask, save)
        if save:
            c = _Ctx()
            c.fus, c.A, c.layers, c.mode, c.row_mask = fus, A, layers, mode, row_mask
            c.batch, c.seq, c.modalities = batch, seq, modalities
            c.lens = [t.shape[1] for t in inputs]
            c.dtypes = [t.dtype for t in inputs]
            ctx.c = c
        ctx.n_in = n_in
        # second output: the last input as the layers saw it (a copy: x itself is saved for the backward)
        return y.view(batch, seq, dim), x.view(batch, seq, dim)[:, seq - inputs[-1].shape[1]:].clone()

    @staticmethod
    def backward(ctx, dy, d_last):
        c: _Ctx = ctx.c
        fus, A = c.fus, c.A
        dim = fus.dim
        dx = fusion_stack_bwd(fus, A, fus.compute_dtype, dy.contiguous().view(-1, dim).float(), c.layers, c.batch,
                              c.seq, c.mode, c.row_mask)
        dxv = dx.view(c.batch, c.seq, dim)
        if d_last is not None:                             # a reader of the last input beside the layers (the mask head)
            ops.rows_accumulate(d_last.float().contiguous(), dxv[:, c.seq - c.lens[-1]:])
        d_type = A.grad_view(fus.token_type_embeddings.weight)
        d_type.zero_()
        d_inputs, off = [], 0
        for mod, n, dt in zip(c.modalities, c.lens, c.dtypes):
            seg = dxv[:, off:off + n]
            d_type[mod] += seg.sum(dim=(0, 1))
            d_inputs.append(seg.to(dt))
            off += n
        ctx.c = None
        n_par = len(ctx.needs_input_grad) - 4 - ctx.n_in
        return (None, None, None, None) + tuple(d_inputs) + tuple(None for _ in range(n_par))


class _DecoderHeadsFn(torch.autograd.Function):
    """The decoders of PickPlaceConvDecoder.forward on their own (models/pickplace.py:66-123); `last_in` (mask head
    only): the fusion model's last input, [B, n_patches+1, dim]."""

    @staticmethod
    def forward(ctx, pp, fused, last_in, *params):
        batch, n, dim = fused.shape
        if n != pp.num_patches + 1:
            raise RuntimeError(f"expected {pp.num_patches + 1} tokens from the fusion model, got {n}")
        tok = fused.float().contiguous()
        extra, mtok = None, None
        if pp.has_mask:
            if last_in is None or tuple(last_in.shape) != (batch, n, dim):
                raise RuntimeError("compute_mask=True: the mask head needs the fusion model's last input [B, n, dim]")
            mtok = last_in.float().contiguous()
            heat, w_eff, rows_ws, extra = decoder_heads_fwd(pp, tok, n * dim, batch, mtok, n * dim)
        else:
            heat, w_eff, rows_ws = decoder_heads_fwd(pp, tok, n * dim, batch)
        if any(ctx.needs_input_grad):
            c = _Ctx()
            c.pp, c.tok, c.heat, c.w_eff, c.rows_ws, c.batch, c.dtype = pp, tok, heat, w_eff, rows_ws, batch, fused.dtype
            c.extra, c.mtok = extra, mtok
            ctx.c = c
        return decoder_outputs(pp, heat, extra)

    @staticmethod
    def backward(ctx, *d_heats):
        c: _Ctx = ctx.c
        pp = c.pp
        A = component_arena(pp)
        d_tok = torch.empty_like(c.tok)
        n, dim = c.tok.shape[1], c.tok.shape[2]
        d_last = None
        if c.extra is not None:
            done, d_last = decoder_heads_bwd(pp, A, c.tok, n * dim, c.heat, c.w_eff, c.rows_ws, d_heats, d_tok, n * dim,
                                             c.batch, None, c.extra, c.mtok, n * dim)
        else:
            done = decoder_heads_bwd(pp, A, c.tok, n * dim, c.heat, c.w_eff, c.rows_ws, d_heats, d_tok, n * dim, c.batch)
        torch.cuda.current_stream().wait_event(done)
        ctx.c = None
        return (None, d_tok.to(c.dtype), d_last) + tuple(None for _ in range(len(ctx.needs_input_grad) - 3))


class RGBTailFn(torch.autograd.Function):
    """project + token assembly + fusion transformer + decoders for RGBOnly
    (models/rgb_clip.py:52-73 -> fusion.py:35-64 -> pickplace.py:59-124)."""

    @staticmethod
    def forward(ctx, model, text_feat, vis_feat, *params):
        # `params` only tie the graph to the trainable parameters; values are read from the arena.
        pp: PickPlaceConvDecoder = model.pick_place
        fus: ConcatTransformer = pp.fusion
        A: ParamArena = model.trainable_arena()
        cdt = model.compute_dtype
        save = any(ctx.needs_input_grad)      # False under no_grad (inference / get_action)
        dim = pp.dim
        n_rgb = pp.num_patches + 1
        batch = vis_feat.shape[0] // n_rgb
        n_text = text_feat.shape[0] // batch
        seq = 1 + n_text + n_rgb
        # K8: the fusion input X [B, 1 + n_text + n_rgb, dim] is written in place by its two
        # producers -- no rgb_feat round trip through HBM, no separate assembly pass over the image rows:
        #   rows 0 .. n_text      text_token / frozen text features + text_pos + type[0]   (bf_segment_assemble)
        #   rows 1 + n_text ..    project(vis_feat) + bias + (rgb_pos + type[1]): the GEMM's epilogue adds the
        #                         per-token table as a residual indexed m % n_rgb and remaps row m of the
        #                         [B * n_rgb] problem to row (m / n_rgb) * seq + 1 + n_text + m % n_rgb
        # (models/rgb_clip.py:52-66 + models/fusion.py:45-56).
        type_w = fus.token_type_embeddings.weight.data
        x = torch.empty(batch * seq, dim, dtype=torch.float32, device=vis_feat.device)
        # nn.Dropout on the embeddings (rgb_clip.py:56,65; training, p > 0): the mask sits BETWEEN the position add and the
        # type-embedding add, so the two are separated -- the rows are assembled without the type vector, masked in place
        # (bf_dropout_rows), and the type vector is added behind the mask by the same kernel
        p_t, p_r = float(getattr(model, "text_dropout", 0.0)), float(getattr(model, "rgb_dropout", 0.0))
        emb_drop = None
        if save and model.training and (p_t > 0.0 or p_r > 0.0):
            base = int(torch.empty((), dtype=torch.int64).random_().item())
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                base ^= (torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
            emb_drop = dict(p_text=p_t, p_rgb=p_r, text=base & (2 ** 63 - 1), rgb=(base + 1) & (2 ** 63 - 1))
            model.last_dropout_seeds = dict(emb_drop)
        zero = torch.zeros(dim, dtype=torch.float32, device=vis_feat.device) if emb_drop is not None else None
        ops.segment_assemble(text_feat, n_text * dim, model.text_token.data.view(-1),
                             model.text_pos_embedding.data.view(-1, dim)[:1 + n_text],
                             type_w[0] if emb_drop is None else zero, x, batch, n_text, seq, 0)
        rgb_table = model.rgb_pos_embedding.data.view(-1, dim) + (type_w[1] if emb_drop is None else zero)   # [n_rgb, dim], 400 KB of glue
        ops.gemm(vis_feat, A.operand(model.project.weight, cdt), bias=model.project.bias.data,
                 residual=rgb_table, resid_mod=n_rgb, out=x, rows_in=n_rgb, rows_out=seq, row_off=1 + n_text)
        if emb_drop is not None:
            ops.dropout_rows(x, type_w[0].contiguous(), batch, seq, 0, 1 + n_text, p_t, emb_drop["text"])
            ops.dropout_rows(x, type_w[1].contiguous(), batch, seq, 1 + n_text, n_rgb, p_r, emb_drop["rgb"])
        outs, saved = fusion_decoder_fwd(pp, A, cdt, x, batch, seq, L.MASK_NONE, None, save)
        if save:
            c = _Ctx()
            c.model, c.saved, c.vis_feat = model, saved, vis_feat
            c.batch, c.n_text, c.n_rgb = batch, n_text, n_rgb
            c.emb_drop = emb_drop
            ctx.c = c
        return outs

    @staticmethod
    def backward(ctx, *d_heats):
        c: _Ctx = ctx.c
        model = c.model
        pp: PickPlaceConvDecoder = model.pick_place
        A: ParamArena = model.trainable_arena()
        A.begin_backward()
        cdt = model.compute_dtype
        g = A.grad_view
        dim = pp.dim
        gs = scaler_of(model)                                  # fp16: loss scale (models/gradscale.py)
        ready = gs.ready if gs is not None else model._grad_ready
        dx = fusion_decoder_bwd(pp, A, cdt, c.saved, d_heats, ready, gs.begin(d_heats, c.saved) if gs is not None else None)
        g_type = g(pp.fusion.token_type_embeddings.weight)
        type_true = None
        if c.emb_drop is not None:
            # the type embeddings were added BEHIND the masks: their gradient sums the gradient as it is; everything in
            # front of the masks (tokens, positions, project) takes the masked, rescaled gradient
            ops.fusion_assemble_bwd(dx, cdt, g(model.text_token).view(-1), g(model.text_pos_embedding).view(-1, dim),
                                    g(model.rgb_pos_embedding).view(-1, dim), g_type, c.batch, c.n_text, c.n_rgb)
            type_true = g_type.clone()
            dx = dx.clone()
            seq = 1 + c.n_text + c.n_rgb
            ops.dropout_rows(dx, None, c.batch, seq, 0, 1 + c.n_text, c.emb_drop["p_text"], c.emb_drop["text"])
            ops.dropout_rows(dx, None, c.batch, seq, 1 + c.n_text, c.n_rgb, c.emb_drop["p_rgb"], c.emb_drop["rgb"])
        d_rgb = ops.fusion_assemble_bwd(dx, cdt, g(model.text_token).view(-1),
                                        g(model.text_pos_embedding).view(-1, dim),
                                        g(model.rgb_pos_embedding).view(-1, dim), g_type, c.batch, c.n_text, c.n_rgb)
        ops.gemm(d_rgb, c.vis_feat, trans_a=True, b_kn=True, out=g(model.project.weight))
        # d project.bias = the sum of d rgb_feat over every image token of the batch = the gradient of the image
        # modality's type embedding, which the assembly backward has just reduced in fp32
        g(model.project.bias).copy_(g_type[1])
        if type_true is not None:
            g_type.copy_(type_true)
        ready(-1)
        ctx.c = None
        # gradients were written into the arena (and attached as .grad) by the kernels
        return (None, None, None) + tuple(None for _ in range(len(ctx.needs_input_grad) - 3))


class SegmentTailFn(torch.autograd.Function):
    """Fusion + decoders for models whose fusion input is a list of [lead token | frozen tower
    tokens] segments (SigLip / SiglipSequential, models/siglip.py:56-70,85-137).

    `segments`: list of dicts  {src: fp32 [B, n, dim] view (frozen tower output), bstride, n,
    lead: Parameter (image/text token), pos: (Parameter, first_row) or None, modality: int}."""

    @staticmethod
    def forward(ctx, model, segments, row_mask, *params):
        pp: PickPlaceConvDecoder = model.pick_place
        fus: ConcatTransformer = pp.fusion
        A: ParamArena = model.trainable_arena()
        cdt = model.compute_dtype
        save = any(ctx.needs_input_grad)
        dim = pp.dim
        batch = segments[0]["batch"]
        seq = sum(1 + sg["n"] for sg in segments)
        dev = segments[0]["src"].device
        x = torch.empty(batch * seq, dim, dtype=torch.float32, device=dev)
        type_w = fus.token_type_embeddings.weight.data
        off = 0
        for sg in segments:
            pos = None
            if sg["pos"] is not None:
                prm, row0 = sg["pos"]
                pos = prm.data.view(-1, dim)[row0:row0 + 1 + sg["n"]]
            ops.segment_assemble(sg["src"], sg["bstride"], sg["lead"].data.view(-1), pos, type_w[sg["modality"]],
                                 x, batch, sg["n"], seq, off)
            sg["row_off"] = off
            off += 1 + sg["n"]
        mode = L.MASK_QUERY_ROW if row_mask is not None else L.MASK_NONE
        outs, saved = fusion_decoder_fwd(pp, A, cdt, x, batch, seq, mode, row_mask, save)
        if save:
            c = _Ctx()
            c.model, c.saved, c.batch, c.seq = model, saved, batch, seq
            c.segments = [{k: v for k, v in sg.items() if k != "src"} for sg in segments]
            ctx.c = c
        return outs

    @staticmethod
    def backward(ctx, *d_heats):
        c: _Ctx = ctx.c
        model = c.model
        pp: PickPlaceConvDecoder = model.pick_place
        A: ParamArena = model.trainable_arena()
        A.begin_backward()
        cdt = model.compute_dtype
        g = A.grad_view
        dim = pp.dim
        gs = scaler_of(model)                                  # fp16: loss scale (models/gradscale.py)
        ready = gs.ready if gs is not None else model._grad_ready
        dx = fusion_decoder_bwd(pp, A, cdt, c.saved, d_heats, ready, gs.begin(d_heats, c.saved) if gs is not None else None)
        if dx.dtype != torch.float32:                          # fp16 mode's 16-bit gradient stream: the segment kernels
            dx = ops.cast(dx, torch.float32)                   # and the towers' backward take fp32
        d_type = g(pp.fusion.token_type_embeddings.weight)
        d_type.zero_()
        leads = {id(sg["lead"]): sg["lead"] for sg in c.segments}
        for prm in leads.values():
            g(prm).zero_()
        for sg in c.segments:
            d_pos = None
            if sg["pos"] is not None:
                prm, row0 = sg["pos"]
                d_pos = g(prm).view(-1, dim)[row0:row0 + 1 + sg["n"]]
            ops.segment_assemble_bwd(dx, g(sg["lead"]).view(-1), d_pos, d_type[sg["modality"]], c.batch, sg["n"],
                                     c.seq, sg["row_off"])
        model.tower_backward(dx, c.segments, c.batch, c.seq)    # LoRA adapters of the towers (no-op when frozen)
        ready(-1)
        ctx.c = None
        return (None, None, None) + tuple(None for _ in range(len(ctx.needs_input_grad) - 3))

"""Tensor-level wrappers over the C ABI: shape/dtype/device checks, output
allocation through PyTorch's caching allocator, launch on the current HIP stream.

PyTorch is plumbing here (device memory + streams); every kernel launched from
this file lives in bifold_amd/csrc/*.hip.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

from . import lib as L
from .lib import check, lib

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}
_TORCH = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16}
HALF = (torch.bfloat16, torch.float16)        # the 16-bit MFMA operand types (one per call)
_TAG = {L.F32: "f32", L.BF16: "bf16", L.F16: "f16"}


class KernelTimer:
    """Optional HIP-event timing of the matrix kernels, on the stream they are launched on
    (bench.py turns it on for the timed region to price the dominant kernel against the MFMA
    roofline).  records: (label, algorithmic FLOPs, start event, end event)."""

    def __init__(self, per_shape: bool = False, only_prefix: Optional[str] = None):
        self.records = []
        self.per_shape = per_shape
        self.only_prefix = only_prefix        # e.g. "gemm_bf16_nt": time just the dominant kernel

    def wants(self, label: str) -> bool:
        return self.only_prefix is None or label.startswith(self.only_prefix)

    def summary(self):
        out = {}
        for label, flops, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = out.setdefault(label, {"launches": 0, "ms": 0.0, "flops": 0.0})
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += flops
        for d in out.values():
            d["avg_us"] = 1e3 * d["ms"] / d["launches"]
            d["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        return out


PROFILER: Optional[KernelTimer] = None


class _Span:
    __slots__ = ("label", "flops", "e0")

    def __init__(self, label, flops):
        self.label, self.flops = label, flops
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def close(self):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILER.records.append((self.label, self.flops, self.e0, e1))


def torch_dtype(code: int) -> torch.dtype:
    return _TORCH[code]


def dtype_code(t) -> int:
    dt = t if isinstance(t, torch.dtype) else t.dtype
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"bifold_hip supports float32 / bfloat16 / float16 tensors, got {dt}") from None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, contiguous: bool = True):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def _rows2d(t: torch.Tensor, name: str):
    """2-D tensor with unit column stride; returns (tensor, rows, cols, ld)."""
    _dev(t, name, contiguous=False)
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{name} must be 2-D with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    return t, t.shape[0], t.shape[1], t.stride(0)


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype: torch.dtype,
                  save_stats: bool = False, eps: float = 1e-5, out: Optional[torch.Tensor] = None):
    _dev(x, "x"); _dev(gamma, "gamma"); _dev(beta, "beta")
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = out if out is not None else torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.bf_layernorm_fwd(_ptr(x), dtype_code(x), _ptr(gamma), _ptr(beta), _ptr(y), dtype_code(y),
                               _ptr(mean), _ptr(rstd), rows, cols, eps, _stream()), "bf_layernorm_fwd")
    return y, mean, rstd


def add_layernorm_fwd(x: torch.Tensor, delta: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                      out_dtype: torch.dtype, x_out: Optional[torch.Tensor] = None, save_stats: bool = False,
                      eps: float = 1e-5, delta2: Optional[torch.Tensor] = None, write_x: bool = True):
    """x_out = x + delta (fp32 residual stream + the branch's 16-bit output), y = LN(x_out).  `x_out=None` updates x in
    place.  `delta2`: a second branch, x_out = (x + delta) + delta2; `write_x=False`: only y is produced, the stream is
    left as it is (bf_add2_layernorm_fwd: the frozen towers write their stream once per block).
    Returns (y, mean, rstd, x_out)."""
    _dev(x, "x"); _dev(delta, "delta"); _dev(gamma, "gamma"); _dev(beta, "beta")
    if (x.dtype != torch.float32 or delta.dtype not in HALF or x.shape != delta.shape
            or (out_dtype in HALF and out_dtype != delta.dtype)):
        raise TypeError("add_layernorm_fwd: x fp32, delta a 16-bit tensor of the same shape, a 16-bit y of delta's type")
    if delta2 is not None:
        _dev(delta2, "delta2")
        if delta2.dtype != delta.dtype or delta2.shape != delta.shape or not delta2.is_contiguous():
            raise TypeError("add_layernorm_fwd: delta2 has delta's dtype and shape")
    cols = x.shape[-1]
    rows = x.numel() // cols
    xo = (x if x_out is None else x_out) if write_x else None
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.bf_add2_layernorm_fwd(_ptr(x), _ptr(delta), _ptr(delta2), dtype_code(delta), _ptr(gamma), _ptr(beta), _ptr(xo),
                                    _ptr(y), dtype_code(y), _ptr(mean), _ptr(rstd), rows, cols, eps, _stream()),
          "bf_add2_layernorm_fwd")
    return y, mean, rstd, xo


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor,
                  rstd: torch.Tensor, dres: Optional[torch.Tensor], cdt: Optional[torch.dtype],
                  dgamma_out: torch.Tensor, dbeta_out: torch.Tensor,
                  dres_sum_out: Optional[torch.Tensor] = None, dx_sum_out: Optional[torch.Tensor] = None,
                  stream16: bool = False):
    """Returns (dx fp32, dx_cdt or None); writes dgamma/dbeta (and, when asked, the column sums
    of dres and of dx) into the given fp32 vectors.  `stream16`: the 16-bit gradient stream -- `dres` (if any) is a
    `cdt` tensor, no fp32 dx is produced (returns (None, dx_cdt))."""
    _dev(dy, "dy"); _dev(x, "x")
    cols = x.shape[-1]
    rows = x.numel() // cols
    if stream16:
        if cdt not in HALF or dy.dtype != cdt or (dres is not None and dres.dtype != cdt):
            raise TypeError("layernorm_bwd(stream16=True): dy, dres and the result share one 16-bit type")
    elif dres is not None and dres.dtype != torch.float32:
        raise TypeError("layernorm_bwd: dres is the fp32 residual gradient (16-bit only with stream16=True)")
    dx = None if stream16 else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dx_c = torch.empty(x.shape, dtype=cdt, device=x.device) if (cdt is not None and cdt != torch.float32) else None
    nparts = lib.bf_layernorm_bwd_parts(rows)
    want = dres_sum_out is not None
    parts = torch.empty(4 if want else 2, nparts, cols, dtype=torch.float32, device=x.device)
    check(lib.bf_layernorm_bwd(_ptr(dy), dtype_code(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd),
                               _ptr(dres), _ptr(dx), _ptr(dx_c), dtype_code(dx_c) if dx_c is not None else L.F32,
                               _ptr(parts[0]), _ptr(parts[1]), _ptr(parts[2]) if want else None,
                               _ptr(parts[3]) if want else None, rows, cols, _stream()), "bf_layernorm_bwd")
    check(lib.bf_colsum_multi(_ptr(parts), 4 if want else 2, nparts, cols, _ptr(dgamma_out), _ptr(dbeta_out),
                              _ptr(dres_sum_out) if want else None, _ptr(dx_sum_out) if want else None,
                              _stream()), "bf_colsum_multi")
    return dx, dx_c


# ----------------------------------------------------------------------------- GEMM
_VARIANT_TAG = {8: "p8", 4: "k32", 0: "bk64", 1: "reg"}


def pick_split_k(m: int, n: int, k: int, deep_tn: bool = False) -> int:
    """Split the contraction when the output alone cannot fill the chip (the dW = dY^T X GEMMs:
    16-64 output tiles, 70 k tokens deep)."""
    if deep_tn and k >= 16384:
        # bf16 dW GEMMs run on the persistent 256x256 kernel: one (tile, split) item per CU in a
        # single round (bf_gemm picks that kernel when tiles * split lands in (3/4, 1] x 256)
        t256 = ((m + 255) // 256) * ((n + 255) // 256)
        if t256 <= 128:
            return max(1, min(64, 256 // t256))
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    if tiles >= 256 or k < 1024:
        return 1
    # measured at B = 256: ~768 blocks beat a "full" 1024 (fewer fp32 partial slabs to write + reduce)
    return max(1, min(64, 768 // tiles, k // 512))


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a: bool = False, b_kn: bool = False,
         bias: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, want_pre: bool = False,
         aux: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         resid_mod: int = 0, out: Optional[torch.Tensor] = None,
         out_dtype: Optional[torch.dtype] = None, rows_in: int = 0, rows_out: int = 0,
         row_off: int = 0, accumulate: bool = False, want_bf16_copy: bool = False,
         alpha: float = 1.0, split_k: Optional[int] = None, colsum_out: Optional[torch.Tensor] = None,
         lr_u: Optional[torch.Tensor] = None, lr_w: Optional[torch.Tensor] = None):
    """C = epilogue(alpha * op(A) op(B)); see include/bifold_hip.h for the epilogue order.
    Returns C, or (C, pre) / (C, copy) when asked for.  lr_u [m, r] / lr_w [n, r] (r = 16 or 32, the operands' dtype):
    a low-rank addend of the product, for calls gemm_lowrank_ok accepts."""
    a, ar, ac, lda = _rows2d(a, "A")
    b, br, bc, ldb = _rows2d(b, "B")
    if a.dtype != b.dtype:
        raise TypeError(f"A and B must share a dtype ({a.dtype} vs {b.dtype})")
    m, k = (ac, ar) if trans_a else (ar, ac)
    n, kb = (bc, br) if b_kn else (br, bc)
    if k != kb:
        raise RuntimeError(f"GEMM inner dimensions differ: {k} vs {kb}")
    cdt = dtype_code(a)
    if out is None:
        od = out_dtype or a.dtype
        if rows_in:
            raise RuntimeError("row remap needs an explicit `out`")
        out = torch.empty(m, n, dtype=od, device=a.device)
    out2, orows, ocols, ldc = _rows2d(out, "C")
    g = L.GemmArgs()
    g.a, g.b, g.c = a.data_ptr(), b.data_ptr(), out.data_ptr()
    g.m, g.n, g.k = m, n, k
    g.lda, g.ldb, g.ldc = lda, ldb, ldc
    g.cdt, g.c_dtype = cdt, dtype_code(out)
    g.trans_a, g.b_kn = int(trans_a), int(b_kn)
    g.alpha = alpha
    g.bias = _ptr(bias)
    g.act = act
    pre = None
    if want_pre:
        pre = torch.empty(out.shape, dtype=a.dtype, device=a.device)
        if pre.stride(0) != ldc:
            raise RuntimeError("pre-activation output needs the same leading dimension as C")
        g.pre = pre.data_ptr()
    if aux is not None:
        aux, _, _, ld_aux = _rows2d(aux, "aux")
        g.aux, g.ld_aux = aux.data_ptr(), ld_aux
    if residual is not None:
        residual, _, _, ld_res = _rows2d(residual, "residual")
        if residual.dtype != torch.float32:
            raise TypeError("residual must be float32")
        g.residual, g.ld_res = residual.data_ptr(), ld_res
    g.resid_mod = resid_mod
    g.rows_in, g.rows_out, g.row_off = rows_in, rows_out, row_off
    g.accumulate = int(accumulate)
    copy = None
    if want_bf16_copy:
        copy = torch.empty(out.shape, dtype=a.dtype if a.dtype in HALF else torch.bfloat16, device=a.device)
        g.c2 = copy.data_ptr()
    if colsum_out is not None:
        if colsum_out.dtype != torch.float32 or colsum_out.numel() != n or not colsum_out.is_contiguous():
            raise RuntimeError("colsum_out must be a contiguous float32 [n] tensor")
        g.colsum_out = colsum_out.data_ptr()
    if lr_u is not None:
        if lr_w is None or lr_u.dtype != a.dtype or lr_w.dtype != a.dtype or lr_u.dim() != 2 or lr_w.dim() != 2 \
                or lr_u.shape[0] != m or lr_w.shape[0] != n or lr_u.shape[1] != lr_w.shape[1] \
                or not lr_u.is_contiguous() or not lr_w.is_contiguous():
            raise RuntimeError("low-rank addend: lr_u [m, r] and lr_w [n, r], contiguous, in the operands' dtype")
        _dev(lr_u, "lr_u"), _dev(lr_w, "lr_w")
        g.lr_u, g.lr_w, g.lr_k = lr_u.data_ptr(), lr_w.data_ptr(), lr_u.shape[1]
    sk = (pick_split_k(m, n, k, deep_tn=(trans_a and b_kn and a.dtype in HALF))
          if split_k is None else split_k)
    ws = None
    if sk > 1:
        ws = torch.empty(sk * m * n, dtype=torch.float32, device=a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    g.split_k = sk
    if sk <= 1 and cdt != L.F32 and m >= 16384 and not os.environ.get("BIFOLD_TEST_NO_TAIL_WS"):
        tail = lib.bf_gemm_tail_workspace_bytes(C.byref(g))
        if tail:
            ws = torch.empty(tail // 4, dtype=torch.float32, device=a.device)
            g.workspace, g.workspace_bytes = ws.data_ptr(), tail
    span = None
    label = f"gemm_{_TAG[cdt]}_{'t' if trans_a else 'n'}{'n' if b_kn else 't'}"
    if PROFILER is not None and PROFILER.wants(label):
        if PROFILER.per_shape:
            extras = "".join(t for t, on in (("+bias", bias is not None), (f"+act{act}", act != 0),
                                              ("+pre", want_pre), ("+res", residual is not None),
                                              ("+f32out", out.dtype == torch.float32)) if on)
            label += f" {m}x{n}x{k}{extras}"
        span = _Span(label, 2.0 * m * n * k)
    check(lib.bf_gemm(C.byref(g), _stream()), "bf_gemm")
    if span is not None:
        if cdt != L.F32:           # name the device kernel that ran: gemm_bf16_nt[p8] / [k32] / [bk64] / [reg]
            span.label = span.label.replace(" ", f"[{_VARIANT_TAG.get(lib.bf_gemm_last_variant(), '?')}] ", 1) \
                if " " in span.label else span.label + f"[{_VARIANT_TAG.get(lib.bf_gemm_last_variant(), '?')}]"
        span.close()
    if want_pre:
        return out, pre
    if want_bf16_copy:
        return out, copy
    return out


def gemm_tn_grouped_qualifies(k: int, dtype: torch.dtype) -> bool:
    """Will a group of dW = dY^T X problems over `k` rows (tokens) be taken by the grouped launch at all?  The library's
    own answer (bf_gemm_tn_grouped_workspace_bytes on a one-tile probe problem): callers decide ONCE per backward
    whether to defer a layer's weight gradients to the end of the layer or to issue them as they come."""
    if dtype not in HALF:
        return False
    arr = (L.GemmTnProblem * 1)()
    arr[0].m, arr[0].n, arr[0].ld_dy, arr[0].ld_x, arr[0].ldc = 256, 256, 256, 256, 256
    arr[0].dy = arr[0].x = arr[0].c = 4096              # the plan checks alignment only; nothing is dereferenced
    return lib.bf_gemm_tn_grouped_workspace_bytes(arr, 1, int(k), L.F16 if dtype == torch.float16 else L.BF16) > 0


def gemm_tn_grouped(problems, alpha: float = 1.0) -> bool:
    """problems: list of (dy [K, m], x [K, n], out fp32 [m, n]) -- out_p = alpha * dy_p^T x_p for every p in ONE launch of
    the persistent kernel (the weight gradients of a transformer layer).  Returns False, having done nothing, when the
    group does not qualify (bf_gemm_tn_grouped_workspace_bytes == 0): the caller then runs plain GEMMs."""
    count = len(problems)
    if count < 1 or count > 8:
        return False
    arr = (L.GemmTnProblem * count)()
    k = None
    for q, (dy, x, out) in zip(arr, problems):
        dy, kd, m, ld_dy = _rows2d(dy, "dY")
        x, kx, n, ld_x = _rows2d(x, "X")
        out, om, on, ldc = _rows2d(out, "dW")
        if dy.dtype != x.dtype or dy.dtype not in HALF or out.dtype != torch.float32 or kd != kx or (om, on) != (m, n):
            return False
        if k is None:
            k, cdt = kd, dtype_code(dy)
        elif kd != k or dtype_code(dy) != cdt:
            return False
        q.dy, q.x, q.c = dy.data_ptr(), x.data_ptr(), out.data_ptr()
        q.m, q.n, q.ld_dy, q.ld_x, q.ldc = m, n, ld_dy, ld_x, ldc
    need = lib.bf_gemm_tn_grouped_workspace_bytes(arr, count, k, cdt)
    if not need:
        return False
    ws = torch.empty(need // 4, dtype=torch.float32, device=problems[0][0].device)
    span = None
    label = f"gemm_{_TAG[cdt]}_tn[p8g]"
    if PROFILER is not None and PROFILER.wants(label):
        span = _Span(label, sum(2.0 * k * q.m * q.n for q in arr))
    check(lib.bf_gemm_tn_grouped(arr, count, k, cdt, float(alpha), ws.data_ptr(), need, _stream()), "bf_gemm_tn_grouped")
    if span is not None:
        span.close()
    return True


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    x, rows, cols, ld = _rows2d(x, "x")
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    check(lib.bf_colsum(_ptr(x), dtype_code(x), _ptr(out), rows, cols, ld, int(accumulate), _stream()),
          "bf_colsum")
    return out


def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None):
    _dev(src, "src")
    dst = out if out is not None else torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib.bf_cast(_ptr(src), dtype_code(src), _ptr(dst), dtype_code(dst), src.numel(), _stream()),
          "bf_cast")
    return dst


def dropout(x: torch.Tensor, p: float, seed: int, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """y (+)= keep(seed, i) ? x / (1 - p) : 0 with the replayable mask of bf_dropout."""
    _dev(x, "x")
    y = out if out is not None else torch.empty_like(x)
    if accumulate and out is None:
        raise RuntimeError("accumulate needs an `out` to add into")
    _dev(y, "out")
    if y.shape != x.shape or y.dtype != x.dtype:
        raise RuntimeError("dropout: out must match x")
    check(lib.bf_dropout(_ptr(x), _ptr(y), dtype_code(x), x.numel(), float(p), int(seed) & (2 ** 64 - 1),
                         int(accumulate), _stream()), "bf_dropout")
    return y


def gemm_lowrank_ok(m: int, n: int, k: int, dtype: torch.dtype, lr_k: int, act: int = L.ACT_NONE) -> bool:
    """Would ops.gemm(a [m, k], b [n, k], bias=..., act=act) in `dtype` carry a low-rank addend of width lr_k?
    (Dense operands and output; the library evaluates the same predicates as its dispatch.)"""
    if dtype not in HALF:
        return False
    g = L.GemmArgs()
    g.m, g.n, g.k, g.lda, g.ldb, g.ldc = m, n, k, k, k, n
    g.cdt = g.c_dtype = L.F16 if dtype == torch.float16 else L.BF16
    g.alpha, g.act, g.lr_k = 1.0, act, lr_k
    g.split_k = pick_split_k(m, n, k, deep_tn=False)
    return bool(lib.bf_gemm_lowrank_ok(C.byref(g)))


def _lora_slots(n_slots: int, rank: int, col_off, p: float, seeds) -> "L.LoraSlots":
    sl = L.LoraSlots()
    sl.n_slots, sl.rank, sl.p = int(n_slots), int(rank), float(p)
    for s in range(n_slots):
        sl.col_off[s] = int(col_off[s])
        sl.seed[s] = (int(seeds[s]) & (2 ** 64 - 1)) if seeds is not None else 0
    return sl


def lora_supported(dtype: torch.dtype, width: int, rank: int, n_slots: int) -> bool:
    """Shapes the bf_lora_* kernels take (the others run the adapter branch as dropout + GEMMs)."""
    return dtype in HALF and rank in (8, 16) and n_slots * rank <= 32 and 1 <= n_slots <= 3 and width % 16 == 0 \
        and (n_slots * rank + 1) * (2 * width + 16) <= 65536          # bf_lora_down keeps the stacked A rows in LDS


def lora_down(x: torch.Tensor, col_off, wt: torch.Tensor, rank: int, width: int, p: float = 0.0, seeds=None,
              alpha: float = 1.0, want_half: bool = False):
    """u fp32 [rows, S * rank]: u[:, slot s] = alpha (mask_s . x[:, col_off[s] : col_off[s] + width]) wt[slot s rows]^T.
    x: 16-bit [rows, ld] (ld = x.shape[1]); wt: same dtype [S * rank, width].  want_half: also return the same values in
    x's dtype (the lr_u operand of ops.gemm)."""
    _dev(x, "x"), _dev(wt, "wt")
    n = len(col_off)
    if x.dim() != 2 or wt.dtype != x.dtype or tuple(wt.shape) != (n * rank, width) or not wt.is_contiguous() \
            or x.stride(1) != 1:
        raise RuntimeError("lora_down: x [rows, ld] and wt [S * rank, width] of one 16-bit type expected")
    rows = x.shape[0]
    u = torch.empty(rows, n * rank, dtype=torch.float32, device=x.device)
    uh = torch.empty(rows, n * rank, dtype=x.dtype, device=x.device) if want_half else None
    sl = _lora_slots(n, rank, col_off, p, seeds)
    check(lib.bf_lora_down(_ptr(x), x.stride(0), _ptr(wt), _ptr(u), _ptr(uh), rows, width, C.byref(sl), float(alpha),
                           dtype_code(x), _stream()), "bf_lora_down")
    return (u, uh) if want_half else u


def lora_up(u: torch.Tensor, wt: torch.Tensor, y: torch.Tensor, col_off, rank: int, width: int, p: float = 0.0, seeds=None,
            alpha: float = 1.0) -> torch.Tensor:
    """y[:, col_off[s] + c] += alpha mask_s . (u[:, slot s] wt[slot s rows, c]), in place; y: 16-bit [rows, ld]."""
    _dev(u, "u"), _dev(wt, "wt"), _dev(y, "y")
    n = len(col_off)
    rows = y.shape[0]
    if y.dim() != 2 or wt.dtype != y.dtype or tuple(wt.shape) != (n * rank, width) or not wt.is_contiguous() \
            or y.stride(1) != 1 or u.dtype != torch.float32 or tuple(u.shape) != (rows, n * rank) or not u.is_contiguous():
        raise RuntimeError("lora_up: u fp32 [rows, S * rank], wt [S * rank, width], y [rows, ld] expected")
    sl = _lora_slots(n, rank, col_off, p, seeds)
    check(lib.bf_lora_up(_ptr(u), _ptr(wt), _ptr(y), y.stride(0), rows, width, C.byref(sl), float(alpha), dtype_code(y),
                         _stream()), "bf_lora_up")
    return y


def lora_tn(x: torch.Tensor, col_off, u: torch.Tensor, outs, rank: int, width: int, transposed: bool, p: float = 0.0,
            seeds=None, alpha: float = 1.0) -> None:
    """outs[s] (fp32 [rank, width], or [width, rank] when transposed) = alpha u[:, slot s]^T (mask_s . x[:, col_off[s] : + width])."""
    _dev(x, "x"), _dev(u, "u")
    n = len(col_off)
    rows = x.shape[0]
    if x.dim() != 2 or x.stride(1) != 1 or u.dtype != torch.float32 or tuple(u.shape) != (rows, n * rank) \
            or not u.is_contiguous() or len(outs) != n:
        raise RuntimeError("lora_tn: x [rows, ld], u fp32 [rows, S * rank] and S outputs expected")
    want = (width, rank) if transposed else (rank, width)
    ptrs = (C.c_void_p * n)()
    for s, o in enumerate(outs):
        _dev(o, "out")
        if o.dtype != torch.float32 or tuple(o.shape) != want or not o.is_contiguous():
            raise RuntimeError(f"lora_tn: out[{s}] must be a contiguous float32 {want} tensor")
        ptrs[s] = o.data_ptr()
    sl = _lora_slots(n, rank, col_off, p, seeds)
    elems = int(lib.bf_lora_tn_workspace_elems(rows, width, C.byref(sl)))
    ws = torch.empty(max(elems, 1), dtype=torch.float32, device=x.device)
    check(lib.bf_lora_tn(_ptr(x), x.stride(0), _ptr(u), ptrs, int(transposed), _ptr(ws), elems, rows, width, C.byref(sl),
                         float(alpha), dtype_code(x), _stream()), "bf_lora_tn")


def lora_bwd_input(x: torch.Tensor, du: torch.Tensor, wt: torch.Tensor, y: torch.Tensor, outs, rank: int, width: int,
                   p: float, seeds, alpha: float = 1.0) -> None:
    """lora_tn(x, du -> outs) and lora_up(du, wt -> y) of slots that all sit on the same columns, in one pass
    (n_slots * rank <= 16)."""
    _dev(x, "x"), _dev(du, "du"), _dev(wt, "wt"), _dev(y, "y")
    n = len(outs)
    rows = x.shape[0]
    if x.dim() != 2 or y.dim() != 2 or x.stride(1) != 1 or y.stride(1) != 1 or y.shape[0] != rows or x.dtype != y.dtype \
            or wt.dtype != x.dtype or tuple(wt.shape) != (n * rank, width) or not wt.is_contiguous() \
            or du.dtype != torch.float32 or tuple(du.shape) != (rows, n * rank) or not du.is_contiguous():
        raise RuntimeError("lora_bwd_input: x, y [rows, ld], du fp32 [rows, S * rank], wt [S * rank, width] expected")
    ptrs = (C.c_void_p * n)()
    for s, o in enumerate(outs):
        _dev(o, "out")
        if o.dtype != torch.float32 or tuple(o.shape) != (rank, width) or not o.is_contiguous():
            raise RuntimeError(f"lora_bwd_input: out[{s}] must be a contiguous float32 [{rank}, {width}] tensor")
        ptrs[s] = o.data_ptr()
    sl = _lora_slots(n, rank, [0] * n, p, seeds)
    elems = int(lib.bf_lora_tn_workspace_elems(rows, width, C.byref(sl)))
    ws = torch.empty(max(elems, 1), dtype=torch.float32, device=x.device)
    check(lib.bf_lora_bwd_input(_ptr(x), x.stride(0), _ptr(du), _ptr(wt), _ptr(y), y.stride(0), ptrs, _ptr(ws), elems, rows,
                                width, C.byref(sl), float(alpha), dtype_code(x), _stream()), "bf_lora_bwd_input")


def dropout_rows(x: torch.Tensor, add: Optional[torch.Tensor], batch: int, group_rows: int, row0: int, nrows: int,
                 p: float, seed: int) -> torch.Tensor:
    """In place on rows [row0, row0 + nrows) of every group of x [batch * group_rows, cols]: dropout with the mask of the
    range's own [batch, nrows, cols] tensor, then += add (fp32 [cols] or None)."""
    _dev(x, "x")
    cols = x.shape[-1]
    if x.numel() != batch * group_rows * cols or not x.is_contiguous():
        raise RuntimeError("dropout_rows: x must be a contiguous [batch * group_rows, cols] tensor")
    if add is not None and (add.dtype != torch.float32 or add.numel() != cols or not add.is_contiguous()):
        raise RuntimeError("dropout_rows: add must be a contiguous float32 [cols] vector")
    check(lib.bf_dropout_rows(_ptr(x), dtype_code(x), _ptr(add), batch, group_rows, row0, nrows, cols, float(p),
                              int(seed) & (2 ** 64 - 1), _stream()), "bf_dropout_rows")
    return x


def dropout_mask(shape, p: float, seed: int, device) -> torch.Tensor:
    """The keep decisions of ops.dropout(x, p, seed) for a tensor of `shape`, as uint8 (1 = kept)."""
    mask = torch.empty(shape, dtype=torch.uint8, device=device)
    check(lib.bf_dropout_mask(_ptr(mask), mask.numel(), float(p), int(seed) & (2 ** 64 - 1), _stream()),
          "bf_dropout_mask")
    return mask


# ----------------------------------------------------------------------------- attention
def attention_fwd(qkv: torch.Tensor, batch: int, seq: int, heads: int, head_dim: int, scale: float,
                  mask_mode: int = L.MASK_NONE, row_mask: Optional[torch.Tensor] = None,
                  need_lse: bool = False, dropout: float = 0.0, seed: int = 0):
    """`dropout` > 0: nn.Dropout on the probabilities with the replayable mask of `seed` (bf_attention_dropout_fwd)."""
    _dev(qkv, "qkv")
    if qkv.numel() != batch * seq * 3 * heads * head_dim:
        raise RuntimeError("qkv has the wrong number of elements")
    out = torch.empty(batch * seq, heads * head_dim, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(batch, heads, seq, dtype=torch.float32, device=qkv.device) if need_lse else None
    if row_mask is not None and row_mask.dtype != torch.int32:
        raise TypeError("row_mask must be int32")
    if dropout > 0.0:
        check(lib.bf_attention_dropout_fwd(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(row_mask), batch, seq, heads, head_dim,
                                           scale, mask_mode, dtype_code(qkv), float(dropout), int(seed), _stream()),
              "bf_attention_dropout_fwd")
        return out, lse
    span = None
    if PROFILER is not None and PROFILER.wants("attn_fwd"):
        causal = 0.5 if mask_mode == L.MASK_CAUSAL else 1.0
        span = _Span(f"attn_fwd_{_TAG[dtype_code(qkv)]}_d{head_dim}",
                     4.0 * batch * heads * seq * seq * head_dim * causal)
    check(lib.bf_attention_fwd(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(row_mask), batch, seq, heads,
                               head_dim, scale, mask_mode, dtype_code(qkv), _stream()), "bf_attention_fwd")
    if span is not None:
        span.close()
    return out, lse


def attention_bwd(qkv, out, dout, lse, batch, seq, heads, head_dim, scale, mask_mode=L.MASK_NONE,
                  row_mask=None, dropout: float = 0.0, seed: int = 0):
    _dev(qkv, "qkv"); _dev(out, "out"); _dev(dout, "dout")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(batch, heads, seq, dtype=torch.float32, device=qkv.device)
    if dropout > 0.0:
        check(lib.bf_attention_dropout_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(row_mask), _ptr(dqkv),
                                           _ptr(delta), batch, seq, heads, head_dim, scale, mask_mode, dtype_code(qkv),
                                           float(dropout), int(seed), _stream()), "bf_attention_dropout_bwd")
        return dqkv
    span = None
    if PROFILER is not None and PROFILER.wants("attn_bwd"):
        span = _Span(f"attn_bwd_{_TAG[dtype_code(qkv)]}_d{head_dim}",
                     10.0 * batch * heads * seq * seq * head_dim)
    check(lib.bf_attention_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(row_mask), _ptr(dqkv),
                               _ptr(delta), batch, seq, heads, head_dim, scale, mask_mode,
                               dtype_code(qkv), _stream()), "bf_attention_bwd")
    if span is not None:
        span.close()
    return dqkv


# ----------------------------------------------------------------------------- token plumbing
def patchify(img: torch.Tensor, patch: int, cdt: torch.dtype):
    _dev(img, "rgb")
    if img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3 or img.shape[2] != img.shape[3]:
        raise RuntimeError(f"rgb must be float32 [B,3,S,S], got {img.dtype} {tuple(img.shape)}")
    b, _, s, _ = img.shape
    g = s // patch
    k = 3 * patch * patch
    if patch % 4 == 0 and k % 64 == 0:
        cols = torch.empty(b * g * g, k, dtype=cdt, device=img.device)
        check(lib.bf_patchify(_ptr(img), _ptr(cols), dtype_code(cols), b, s, patch, _stream()), "bf_patchify")
        return cols
    # ViT-L/14 and friends: rows padded with zeros to a multiple of 64 (the weight gets zero columns too)
    ld = patchify_row_length(patch)
    cols = torch.empty(b * g * g, ld, dtype=cdt, device=img.device)
    check(lib.bf_patchify_ld(_ptr(img), _ptr(cols), dtype_code(cols), b, s, patch, ld, _stream()),
          "bf_patchify_ld")
    return cols


def patchify_row_length(patch: int) -> int:
    """Columns of the im2col matrix ops.patchify returns: 3 p^2, rounded up to a multiple of 64 unless
    the fast path applies (the patch-embed weight must be padded to the same K)."""
    k = 3 * patch * patch
    return k if (patch % 4 == 0 and k % 64 == 0) else (k + 63) // 64 * 64


def vision_assemble(patches, cls, pos, gamma, beta, batch: int, n_patch: int, eps: float = 1e-5):
    width = patches.shape[-1]
    x = torch.empty(batch * (n_patch + 1), width, dtype=torch.float32, device=patches.device)
    check(lib.bf_vision_assemble(_ptr(patches), _ptr(cls), _ptr(pos), _ptr(gamma), _ptr(beta), _ptr(x),
                                 batch, n_patch, width, eps, _stream()), "bf_vision_assemble")
    return x


def vision_lnpre_ln1(x, cls, pos0, g0, b0, g1, b1, out_dtype: torch.dtype, batch: int, n_patch: int, eps: float = 1e-5):
    """K1: x fp32 [B*(n+1), W] holds patches + pos in its patch rows (written by the patch-embed GEMM's epilogue);
    finishes it in place (class rows, ln_pre) and returns ln_1(x) in out_dtype."""
    width = x.shape[-1]
    if x.dtype != torch.float32 or not x.is_contiguous() or x.shape[0] != batch * (n_patch + 1):
        raise RuntimeError("vision_lnpre_ln1: x must be a contiguous float32 [B*(n+1), W] tensor")
    h = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib.bf_vision_lnpre_ln1(_ptr(x), _ptr(cls), _ptr(pos0), _ptr(g0), _ptr(b0), _ptr(g1), _ptr(b1), _ptr(h),
                                  dtype_code(h), batch, n_patch, width, eps, _stream()), "bf_vision_lnpre_ln1")
    return h


def embed_tokens(tokens: torch.Tensor, table: torch.Tensor, pos: torch.Tensor, err_flag: torch.Tensor):
    _dev(tokens, "instruction")
    if tokens.dtype != torch.int32:
        raise TypeError("token ids must be int32")
    b, ctx = tokens.shape
    vocab, width = table.shape
    x = torch.empty(b * ctx, width, dtype=torch.float32, device=table.device)
    check(lib.bf_embed_tokens(_ptr(tokens), _ptr(table), _ptr(pos), _ptr(x), b, ctx, width, vocab,
                              _ptr(err_flag), _stream()), "bf_embed_tokens")
    return x


def fusion_assemble(text_feat, rgb_feat, text_token, text_pos, rgb_pos, type_emb, batch, n_text, n_rgb):
    dim = text_feat.shape[-1]
    x = torch.empty(batch * (1 + n_text + n_rgb), dim, dtype=torch.float32, device=text_feat.device)
    check(lib.bf_fusion_assemble(_ptr(text_feat), _ptr(rgb_feat), _ptr(text_token), _ptr(text_pos),
                                 _ptr(rgb_pos), _ptr(type_emb), _ptr(x), batch, n_text, n_rgb, dim,
                                 _stream()), "bf_fusion_assemble")
    return x


def fusion_assemble_bwd(dx, cdt: torch.dtype, d_text_token, d_text_pos, d_rgb_pos, d_type, batch, n_text,
                        n_rgb):
    dim = dx.shape[-1]
    d_rgb = torch.empty(batch * n_rgb, dim, dtype=cdt, device=dx.device)
    check(lib.bf_fusion_assemble_bwd(_ptr(dx), dtype_code(dx), _ptr(d_rgb), dtype_code(d_rgb), _ptr(d_text_token),
                                     _ptr(d_text_pos), _ptr(d_rgb_pos), _ptr(d_type), batch, n_text,
                                     n_rgb, dim, _stream()), "bf_fusion_assemble_bwd")
    return d_rgb


def segment_assemble(src: torch.Tensor, src_bstride: int, lead, pos, type_vec, x, batch: int, n: int,
                     seq_total: int, row_off: int):
    dim = x.shape[-1]
    check(lib.bf_segment_assemble(_ptr(src), src_bstride, _ptr(lead), _ptr(pos), _ptr(type_vec), _ptr(x), batch,
                                  n, seq_total, row_off, dim, _stream()), "bf_segment_assemble")


def segment_assemble_bwd(dx, d_lead, d_pos, d_type, batch: int, n: int, seq_total: int, row_off: int):
    dim = dx.shape[-1]
    tmp = torch.empty(n + 1, dim, dtype=torch.float32, device=dx.device)
    check(lib.bf_segment_assemble_bwd(_ptr(dx), _ptr(d_lead), _ptr(d_pos), _ptr(d_type), _ptr(tmp), batch, n,
                                      seq_total, row_off, dim, _stream()), "bf_segment_assemble_bwd")


# ----------------------------------------------------------------------------- decoder + BCE
def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def decoder_collapse_fwd(ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor], w_eff, b_eff, rows_ws):
    dim = ws[0].shape[1]
    check(lib.bf_decoder_collapse_fwd(_ptr_array(ws), _ptr_array(bs), _ptr(w_eff), _ptr(b_eff),
                                      _ptr(rows_ws), dim, _stream()), "bf_decoder_collapse_fwd")


def decoder_collapse_bwd(ws, bs, rows_ws, g_w_eff, g_b_eff, dws, dbs):
    dim = ws[0].shape[1]
    check(lib.bf_decoder_collapse_bwd(_ptr_array(ws), _ptr_array(bs), _ptr(rows_ws), _ptr(g_w_eff),
                                      _ptr(g_b_eff), _ptr_array(dws), _ptr_array(dbs), dim, _stream()),
          "bf_decoder_collapse_bwd")


def decoder_project(tok: torch.Tensor, tok_bstride: int, w_eff, b_eff, batch, n_heads, n_patch, dim):
    low = torch.empty(n_heads, batch, n_patch, dtype=torch.float32, device=w_eff.device)
    check(lib.bf_decoder_project(_ptr(tok), tok_bstride, _ptr(w_eff), _ptr(b_eff), _ptr(low), batch, n_heads,
                                 n_patch, dim, _stream()), "bf_decoder_project")
    return low


def decoder_upsample_sigmoid(low, U, n_maps: int, grid: int):
    s = grid * 16
    heat = torch.empty(n_maps, s, s, dtype=torch.float32, device=low.device)
    check(lib.bf_decoder_upsample_sigmoid(_ptr(low), _ptr(U), _ptr(heat), n_maps, grid, _stream()),
          "bf_decoder_upsample_sigmoid")
    return heat


def decoder_upsample_sigmoid_bwd(heat, d_heat, U, n_maps: int, grid: int, out: Optional[torch.Tensor] = None):
    g_low = out if out is not None else torch.empty(n_maps, grid * grid, dtype=torch.float32, device=heat.device)
    check(lib.bf_decoder_upsample_sigmoid_bwd(_ptr(heat), _ptr(d_heat), _ptr(U), _ptr(g_low), n_maps, grid,
                                              _stream()), "bf_decoder_upsample_sigmoid_bwd")
    return g_low


def decoder_project_bwd(tok, tok_bstride, w_eff, g_low, d_tok, dtok_bstride, g_w_eff, g_b_eff, batch,
                        n_heads, n_patch, dim):
    ws = torch.empty(lib.bf_decoder_project_bwd_ws(batch, n_heads, dim), dtype=torch.float32,
                     device=w_eff.device)
    check(lib.bf_decoder_project_bwd(_ptr(tok), tok_bstride, _ptr(w_eff), _ptr(g_low), _ptr(d_tok), dtype_code(d_tok),
                                     dtok_bstride, _ptr(g_w_eff), _ptr(g_b_eff), _ptr(ws), batch, n_heads,
                                     n_patch, dim, _stream()), "bf_decoder_project_bwd")


def bce_mean(heat: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean BCE of one head; heat/target [B, ...] fp32 contiguous.  Returns a 1-element tensor."""
    _dev(heat, "heatmap"); _dev(target, "target")
    if heat.shape != target.shape or heat.dtype != torch.float32 or target.dtype != torch.float32:
        raise RuntimeError(f"BCE needs matching float32 tensors, got {tuple(heat.shape)} {heat.dtype} vs "
                           f"{tuple(target.shape)} {target.dtype}")
    batch = heat.shape[0]
    pixels = heat.numel() // batch
    nparts = lib.bf_bce_parts(batch)
    parts = torch.empty(nparts, dtype=torch.float32, device=heat.device)
    loss = torch.empty(1, dtype=torch.float32, device=heat.device)
    check(lib.bf_bce_fwd(_ptr(heat), _ptr(target), _ptr(parts), batch, 1, pixels, pixels, 0, pixels, 0,
                         _stream()), "bf_bce_fwd")
    check(lib.bf_bce_finish(_ptr(parts), _ptr(loss), 1, nparts, 1.0 / heat.numel(), _stream()), "bf_bce_finish")
    return loss


def bce_mean_bwd(heat, target, gscale: torch.Tensor):
    d = torch.empty_like(heat)
    check(lib.bf_bce_bwd(_ptr(heat), _ptr(target), _ptr(gscale), _ptr(d), heat.numel(), 1.0 / heat.numel(),
                         _stream()), "bf_bce_bwd")
    return d


# ----------------------------------------------------------------------------- mask head (compute_mask=True)
def _mask_pair(prob: torch.Tensor, target: torch.Tensor, what: str):
    _dev(prob, what + " probabilities"); _dev(target, what + " target")
    if prob.numel() != target.numel() or prob.dtype != torch.float32 or target.dtype != torch.float32:
        raise RuntimeError(f"{what} needs float32 tensors of one size, got {tuple(prob.shape)} {prob.dtype} vs "
                           f"{tuple(target.shape)} {target.dtype}")
    if prob.shape[0] != target.shape[0]:
        raise RuntimeError(f"{what}: batch sizes differ ({prob.shape[0]} vs {target.shape[0]})")


def mask_mul_fwd(s: torch.Tensor, mask: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """s fp32 [n_heads, B, S, S] (sigmoids of the pick heads) * mask fp32 [B, S, S] -> out like s."""
    _dev(s, "pick heatmaps"); _dev(mask, "mask heatmap")
    if s.dtype != torch.float32 or mask.dtype != torch.float32 or s.shape[1:] != mask.shape:
        raise RuntimeError(f"mask_mul_fwd: fp32 [n, ...] and [...] expected, got {tuple(s.shape)} / {tuple(mask.shape)}")
    out = torch.empty_like(s) if out is None else out
    check(lib.bf_mask_mul_fwd(_ptr(s), _ptr(mask), _ptr(out), s.shape[0], mask.numel(), _stream()), "bf_mask_mul_fwd")
    return out


def mask_mul_bwd(d_outs, s: torch.Tensor, mask: torch.Tensor, d_mask_in: Optional[torch.Tensor], detach: bool,
                 want_d_mask: bool = True):
    """Backward of mask_mul_fwd: (d_s like s, d_mask like mask or None).  `d_outs`: one gradient (or None) per head."""
    n = s.shape[0]
    if len(d_outs) != n:
        raise RuntimeError(f"mask_mul_bwd: {n} heads, {len(d_outs)} gradients")
    keep = []
    arr = (C.c_void_p * n)()
    for i, d in enumerate(d_outs):
        if d is not None:
            d = d.contiguous()
            if d.dtype != torch.float32 or d.shape != mask.shape:
                raise RuntimeError(f"mask_mul_bwd: gradient {i} is {tuple(d.shape)} {d.dtype}")
            keep.append(d)
            arr[i] = d.data_ptr()
    if d_mask_in is not None:
        d_mask_in = d_mask_in.contiguous()
    d_s = torch.empty_like(s)
    d_mask = torch.empty_like(mask) if want_d_mask else None
    check(lib.bf_mask_mul_bwd(arr, _ptr(s), _ptr(mask), _ptr(d_mask_in), _ptr(d_s), _ptr(d_mask), n, mask.numel(),
                              1 if detach else 0, _stream()), "bf_mask_mul_bwd")
    return d_s, d_mask


def rows_accumulate(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[b] += src[b]: src fp32 [B, n, D] contiguous, dst a [B, n, D] view (rows contiguous, any batch stride)."""
    if src.dtype != torch.float32 or not src.is_contiguous() or src.shape != dst.shape or src.dim() != 3:
        raise RuntimeError(f"rows_accumulate: fp32 contiguous [B, n, D] source expected, got {tuple(src.shape)} {src.dtype}")
    if dst.stride(2) != 1 or dst.stride(1) != dst.shape[2]:
        raise RuntimeError("rows_accumulate: the destination's rows must be contiguous")
    check(lib.bf_rows_accumulate(_ptr(src), src.stride(0), _ptr(dst), dtype_code(dst), dst.stride(0), src.shape[0],
                                 src.shape[1] * src.shape[2], _stream()), "bf_rows_accumulate")


def dice_fwd(prob: torch.Tensor, target: torch.Tensor):
    """DiceLoss forward: (loss [1], stats [B, 4] for the backward)."""
    _mask_pair(prob, target, "dice")
    batch = prob.shape[0]
    stats = torch.empty(batch, 4, dtype=torch.float32, device=prob.device)
    loss = torch.empty(1, dtype=torch.float32, device=prob.device)
    check(lib.bf_dice_fwd(_ptr(prob), _ptr(target), _ptr(stats), _ptr(loss), batch, prob.numel() // batch, _stream()),
          "bf_dice_fwd")
    return loss, stats


def dice_bwd(prob_like: torch.Tensor, target: torch.Tensor, stats: torch.Tensor, gscale: torch.Tensor):
    d = torch.empty_like(prob_like)
    batch = prob_like.shape[0]
    check(lib.bf_dice_bwd(_ptr(target), _ptr(stats), _ptr(gscale), _ptr(d), batch, prob_like.numel() // batch, _stream()),
          "bf_dice_bwd")
    return d


def focal_fwd(prob: torch.Tensor, target: torch.Tensor, alpha: float, gamma: float) -> torch.Tensor:
    """FocalLoss forward on [B, rows, cols]: sum / rows.  Returns a 1-element tensor."""
    _mask_pair(prob, target, "focal")
    if prob.dim() != 3:
        raise RuntimeError(f"focal: [B, S, S] probabilities expected, got {tuple(prob.shape)}")
    nparts = lib.bf_bce_parts(prob.shape[0])
    parts = torch.empty(nparts, dtype=torch.float32, device=prob.device)
    loss = torch.empty(1, dtype=torch.float32, device=prob.device)
    check(lib.bf_focal_fwd(_ptr(prob), _ptr(target), _ptr(parts), _ptr(loss), prob.numel(), nparts, 1.0 / prob.shape[1],
                           float(alpha), float(gamma), _stream()), "bf_focal_fwd")
    return loss


def focal_bwd(prob: torch.Tensor, target: torch.Tensor, gscale: torch.Tensor, alpha: float, gamma: float):
    d = torch.empty_like(prob)
    check(lib.bf_focal_bwd(_ptr(prob), _ptr(target), _ptr(gscale), _ptr(d), prob.numel(), 1.0 / prob.shape[1], float(alpha),
                           float(gamma), _stream()), "bf_focal_bwd")
    return d


# ----------------------------------------------------------------------------- optimiser
def adam_step(param, grad, exp_avg, exp_avg_sq, shadow, step, lr, beta1, beta2, eps, weight_decay,
              decoupled, grad_scale=None, skip_flag=None, skipped=None, skipped_known: float = 0.0):
    """`skip_flag` (int32 [1]) non-zero -> the launch changes nothing; `skipped` (fp32 [1]) = steps dropped so far on the
    device, `skipped_known` the part of them `step` already leaves out (bf_adam_step)."""
    if skip_flag is not None and skip_flag.dtype != torch.int32:
        raise TypeError("adam_step: skip_flag int32")
    if skipped is not None and skipped.dtype != torch.float32:
        raise TypeError("adam_step: skipped float32")
    check(lib.bf_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(shadow),
                           dtype_code(shadow) if shadow is not None else L.BF16, param.numel(), step, lr, beta1, beta2, eps, weight_decay, int(decoupled),
                           _ptr(grad_scale), _ptr(skip_flag), _ptr(skipped), float(skipped_known), _stream()), "bf_adam_step")


def scale_check_(x: torch.Tensor, scale, nonfinite: Optional[torch.Tensor] = None) -> None:
    """x *= scale in place (fp32, contiguous); `scale` a number or a device fp32 scalar (the dynamic loss scale);
    `nonfinite` (int32 [1], optional) counts inf / nan results."""
    _dev(x, "x")
    if x.dtype != torch.float32 or (nonfinite is not None and nonfinite.dtype != torch.int32):
        raise TypeError("scale_check_: x float32, nonfinite int32")
    if isinstance(scale, torch.Tensor):
        _dev(scale, "scale")
        if scale.dtype != torch.float32 or scale.numel() < 1:
            raise TypeError("scale_check_: a device scale is one fp32 element")
        check(lib.bf_scale_check_dev(_ptr(x), x.numel(), _ptr(scale), _ptr(nonfinite), _stream()), "bf_scale_check_dev")
    else:
        check(lib.bf_scale_check(_ptr(x), x.numel(), float(scale), _ptr(nonfinite), _stream()), "bf_scale_check")


def gradscale_update_(state: torch.Tensor, flag: torch.Tensor, backoff: float, growth: float, growth_interval: int,
                      min_scale: float, max_scale: float) -> None:
    """End of a guarded step: device-side loss-scale schedule (bf_gradscale_update); clears `flag`."""
    _dev(state, "state")
    _dev(flag, "flag")
    if state.dtype != torch.float32 or state.numel() < 8 or flag.dtype != torch.int32:
        raise TypeError("gradscale_update_: state fp32 [8], flag int32 [1]")
    check(lib.bf_gradscale_update(_ptr(state), _ptr(flag), float(backoff), float(growth), int(growth_interval),
                                  float(min_scale), float(max_scale), _stream()), "bf_gradscale_update")


def sumsq(x: torch.Tensor) -> torch.Tensor:
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib.bf_sumsq(_ptr(x), x.numel(), _ptr(ws), _ptr(out), _stream()), "bf_sumsq")
    return out


def clip_coef(sumsq_t: torch.Tensor, max_norm: float) -> torch.Tensor:
    out = torch.empty(1, dtype=torch.float32, device=sumsq_t.device)
    check(lib.bf_clip_coef(_ptr(sumsq_t), max_norm, _ptr(out), _stream()), "bf_clip_coef")
    return out


def clip_grad_norm_(flat_grad: torch.Tensor, max_norm: float) -> torch.Tensor:
    """In-place torch.nn.utils.clip_grad_norm_ of a flat fp32 gradient arena; returns the total norm
    as a 0-dim device tensor (what torch returns), no host sync."""
    _dev(flat_grad, "grad")
    ws = torch.empty(1026, dtype=torch.float32, device=flat_grad.device)
    norm = torch.empty(1, dtype=torch.float32, device=flat_grad.device)
    check(lib.bf_clip_grad_norm(_ptr(flat_grad), flat_grad.numel(), float(max_norm), _ptr(ws), _ptr(norm),
                                _stream()), "bf_clip_grad_norm")
    return norm.view(())


# ----------------------------------------------------------------------------- either side of the step
def heatmap_pick(heat: torch.Tensor, mask: Optional[torch.Tensor] = None):
    """heat fp32 [n, S, S]; mask [n, S, S] / [n, 1, S, S] (non-zero = allowed) or None ->
    (xy int32 [n, 2] as (x, y), confidence fp32 [n]), all on the device."""
    if heat.dim() != 3 or heat.shape[1] != heat.shape[2] or heat.dtype != torch.float32:
        raise RuntimeError("heat must be fp32 [n, size, size]")
    heat = heat.contiguous()
    _dev(heat, "heat")
    n, size = heat.shape[0], heat.shape[1]
    if mask is not None:
        if mask.numel() != heat.numel():
            raise RuntimeError("mask must have one value per heatmap pixel")
        mask = mask.reshape(n, size, size).to(torch.float32).contiguous()
        _dev(mask, "mask")
    xy = torch.empty(n, 2, dtype=torch.int32, device=heat.device)
    conf = torch.empty(n, dtype=torch.float32, device=heat.device)
    check(lib.bf_heatmap_pick(_ptr(heat), _ptr(mask), _ptr(xy), _ptr(conf), n, size, _stream()),
          "bf_heatmap_pick")
    return xy, conf


def gaussian_targets(points: torch.Tensor, counts: Optional[torch.Tensor], size: int, sigma: float,
                     strategy: str = "gmm") -> torch.Tensor:
    """points fp32 [n, P, 2] (x, y) pixels, counts int32 [n] (None = all P valid) -> fp32 [n, S, S]."""
    if points.dim() != 3 or points.shape[2] != 2 or points.dtype != torch.float32:
        raise RuntimeError("points must be fp32 [n_maps, max_points, 2]")
    codes = {"first": L.TARGETS_FIRST, "gmm": L.TARGETS_GMM, "fit": L.TARGETS_FIT}
    if strategy not in codes:
        raise ValueError(f"Strategy {strategy} not recognized")
    if counts is not None and (counts.dtype != torch.int32 or counts.numel() != points.shape[0]):
        raise RuntimeError("counts must be int32 [n_maps]")
    points = points.contiguous()
    _dev(points, "points")
    n, p = points.shape[0], points.shape[1]
    if not 1 <= p <= L.MAX_TARGET_POINTS:
        raise RuntimeError(f"at most {L.MAX_TARGET_POINTS} points per map (got {p})")
    heat = torch.empty(n, size, size, dtype=torch.float32, device=points.device)
    check(lib.bf_gaussian_targets(_ptr(points), _ptr(counts), _ptr(heat), n, p, size, float(sigma),
                                  codes[strategy], _stream()), "bf_gaussian_targets")
    return heat


def preprocess_rgb(raw: torch.Tensor, mask: Optional[torch.Tensor], mean, std) -> torch.Tensor:
    """raw uint8 [B, S, S, 3] (+ mask [B, S, S] / [B, 1, S, S], 1 = keep) -> fp32 [B, 3, S, S]: mask fill 77, /255,
    (x - mean) / std -- the tensor half of Processor.__call__ for frames already at the model's resolution."""
    if raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[3] != 3 or raw.shape[1] != raw.shape[2]:
        raise RuntimeError(f"raw must be uint8 [B, S, S, 3], got {raw.dtype} {tuple(raw.shape)}")
    raw = raw.contiguous()
    _dev(raw, "raw")
    b, s = raw.shape[0], raw.shape[1]
    if mask is not None:
        if mask.numel() != b * s * s:
            raise RuntimeError("mask must have one value per pixel")
        mask = mask.reshape(b, s, s).to(torch.float32).contiguous()
        _dev(mask, "mask")
    out = torch.empty(b, 3, s, s, dtype=torch.float32, device=raw.device)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.bf_preprocess_rgb(_ptr(raw), _ptr(mask), _ptr(out), b, s, m3, s3, _stream()), "bf_preprocess_rgb")
    return out


def resize_normalize_rgb(raw: torch.Tensor, mask: Optional[torch.Tensor], size: int, mean, std) -> torch.Tensor:
    """raw uint8 [B, H, W, 3] of any resolution (+ mask [B, H, W] / [B, 1, H, W], 1 = keep) -> fp32 [B, 3, size, size]:
    mask fill 77, Resize(size, bicubic, antialias), CenterCrop(size), /255, (x - mean) / std -- the rgb_transform of
    Processor.__call__ (bifold/data/processor.py:87-95) in two launches."""
    if raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[3] != 3:
        raise RuntimeError(f"raw must be uint8 [B, H, W, 3], got {raw.dtype} {tuple(raw.shape)}")
    raw = raw.contiguous()
    _dev(raw, "raw")
    b, h, w = raw.shape[0], raw.shape[1], raw.shape[2]
    if mask is not None:
        if mask.numel() != b * h * w:
            raise RuntimeError("mask must have one value per pixel")
        mask = mask.reshape(b, h, w).to(torch.float32).contiguous()
        _dev(mask, "mask")
    tmp = torch.empty(b, h, size, 3, dtype=torch.uint8, device=raw.device)
    out = torch.empty(b, 3, size, size, dtype=torch.float32, device=raw.device)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.bf_resize_normalize_rgb(_ptr(raw), _ptr(mask), _ptr(tmp), _ptr(out), b, h, w, int(size), m3, s3,
                                      _stream()), "bf_resize_normalize_rgb")
    return out


def depth_frames(depth: torch.Tensor, mask: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
                 standardize: bool = False, thresh: float = 0.1) -> torch.Tensor:
    """The reference's depth transform chain at the model's own resolution (data/processor.py:59-78): DepthScale
    (`shift` [B], one per frame) + MaskDepth (`mask` [B, 1, S, S] / [B, S, S]) + float32 + optionally
    TruncatedDepthStandardization.  depth fp32 [B, 1, S, S] or [B, S, S] -> fp32 of the same shape."""
    _dev(depth, "depth")
    if depth.dtype != torch.float32:
        raise TypeError("depth_frames: depth float32")
    d = depth.contiguous()
    b = d.shape[0]
    pixels = d[0].numel() if b else 1
    for name, t, want in (("mask", mask, d.numel()), ("shift", shift, b)):
        if t is not None and (t.dtype != torch.float32 or t.numel() != want or not t.is_cuda):
            raise TypeError(f"depth_frames: {name} float32 on the device with {want} elements")
    out = torch.empty_like(d)
    check(lib.bf_depth_prepare(_ptr(d), _ptr(mask.contiguous()) if mask is not None else None,
                               _ptr(shift.contiguous()) if shift is not None else None, _ptr(out), b, pixels, _stream()),
          "bf_depth_prepare")
    if standardize:
        lo, hi = int(thresh * pixels), int((1 - thresh) * pixels)        # depth_augmentations.py:12-16, verbatim arithmetic
        check(lib.bf_depth_standardize(_ptr(out), _ptr(out), None, b, pixels, lo, hi, _stream()), "bf_depth_standardize")
    return out


def affine_nearest(x: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
    """x fp32 [N, C, H, W], params fp32 [N, 3] = (angle degrees, dx, dy) on the device -> the warped copy
    (v2.functional.affine with its defaults; Processor._spatial_augmentation)."""
    if x.dim() != 4 or x.dtype != torch.float32:
        raise RuntimeError("affine_nearest needs fp32 [N, C, H, W]")
    x = x.contiguous()
    _dev(x, "x")
    params = params.to(torch.float32).contiguous()
    _dev(params, "params")
    if params.shape != (x.shape[0], 3):
        raise RuntimeError(f"params must be [N, 3] = (angle, dx, dy) per image, got {tuple(params.shape)}")
    out = torch.empty_like(x)
    check(lib.bf_affine_nearest(_ptr(x), _ptr(out), _ptr(params), x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                                _stream()), "bf_affine_nearest")
    return out


# ----------------------------------------------------------------------------- fp8 forward path (frozen towers)
def layernorm_fwd_fp8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """fp32 [rows, cols] -> (e4m3 bytes uint8 [rows, cols], fp32 row scales [rows]); LN(x) ~= bytes * scale."""
    _dev(x, "x"); _dev(gamma, "gamma"); _dev(beta, "beta")
    if x.dtype != torch.float32:
        raise TypeError("layernorm_fwd_fp8 reads the fp32 residual stream")
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty(rows, cols, dtype=torch.uint8, device=x.device)
    scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.bf_layernorm_fwd_fp8(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(scale), rows, cols, eps, _stream()),
          "bf_layernorm_fwd_fp8")
    return y, scale


def quantize_rows_fp8(w: torch.Tensor):
    """[N, K] fp32 / bf16 weights -> (e4m3 bytes uint8 [N, K], fp32 per-row scales [N])."""
    w, rows, cols, ld = _rows2d(w, "w")
    q = torch.empty(rows, cols, dtype=torch.uint8, device=w.device)
    scale = torch.empty(rows, dtype=torch.float32, device=w.device)
    check(lib.bf_quantize_rows_fp8(_ptr(w), dtype_code(w), _ptr(q), _ptr(scale), rows, cols, ld, _stream()),
          "bf_quantize_rows_fp8")
    return q, scale


def gemm_fp8(a8: torch.Tensor, a_scale: torch.Tensor, b8: torch.Tensor, b_scale: torch.Tensor, *,
             bias: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, out_dtype: torch.dtype = torch.bfloat16,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = act((a_scale[m] b_scale[n]) * a8 b8^T + bias) with e4m3 operands (uint8 storage) on the scaled fp8 MFMA."""
    a8, m, k, lda = _rows2d(a8, "A")
    b8, n, kb, ldb = _rows2d(b8, "B")
    if a8.dtype != torch.uint8 or b8.dtype != torch.uint8:
        raise TypeError("gemm_fp8 takes e4m3 operands stored as uint8")
    if k != kb:
        raise RuntimeError(f"GEMM inner dimensions differ: {k} vs {kb}")
    if out is None:
        out = torch.empty(m, n, dtype=out_dtype, device=a8.device)
    out, _, _, ldc = _rows2d(out, "C")
    span = None
    if PROFILER is not None and PROFILER.wants("gemm_fp8_nt"):
        span = _Span("gemm_fp8_nt", 2.0 * m * n * k)
    check(lib.bf_gemm_fp8(_ptr(a8), _ptr(a_scale), _ptr(b8), _ptr(b_scale), _ptr(out), dtype_code(out), _ptr(bias), act,
                          m, n, k, lda, ldb, ldc, _stream()), "bf_gemm_fp8")
    if span is not None:
        span.close()
    return out

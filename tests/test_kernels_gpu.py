"""Kernel-level parity on the MI355X: every C-ABI entry point against a plain PyTorch
fp32/fp64 CPU formula of the same op (and the oracle where it has one).

Tolerances: the exact (fp32) mode is held to 1e-4 relative or tighter, well inside the
north-star's 1e-3; the bf16 mode is compared against the same formula evaluated on the
bf16-ROUNDED operands, so what is left is accumulation order and the final bf16 rounding
(2^-8 relative), stated per test.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


HALF = (torch.bfloat16, torch.float16)      # the two 16-bit compute dtypes: same kernels, other MFMA opcode / rounding


def round_to(t, cdt):
    return t.to(cdt).to(torch.float32)


@pytest.fixture(scope="module")
def ops(gpu):
    from bifold_amd.hip import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def L(gpu):
    from bifold_amd.hip import lib as _lib
    return _lib


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,cols", [(7, 512), (197 * 3, 768), (5, 96), (33, 2048)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_layernorm_fwd(ops, gpu, rows, cols, out_dtype):
    g = torch.Generator().manual_seed(rows * cols)
    x = torch.randn(rows, cols, generator=g) * 3 + 0.7
    w = 1 + 0.2 * torch.randn(cols, generator=g)
    b = 0.2 * torch.randn(cols, generator=g)
    ref = F.layer_norm(x.double(), (cols,), w.double(), b.double(), 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.to(gpu), w.to(gpu), b.to(gpu), out_dtype, save_stats=True)
    tol = 2e-6 if out_dtype == torch.float32 else 2 ** -8
    assert rel_err(y.float(), ref) < tol
    assert rel_err(mean, x.double().mean(1)) < 1e-5
    assert rel_err(rstd, 1 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("rows,cols", [(9, 512), (275 * 2, 768), (40, 96)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_layernorm_bwd(ops, gpu, rows, cols, cdt):
    g = torch.Generator().manual_seed(1 + rows)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.3).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(cols, generator=g)).requires_grad_(True)
    b = (0.2 * torch.randn(cols, generator=g)).requires_grad_(True)
    dy = torch.randn(rows, cols, generator=g)
    dres = torch.randn(rows, cols, generator=g)
    dy_used = round_to(dy, cdt) if cdt in HALF else dy
    y = F.layer_norm(x, (cols,), w, b, 1e-5)
    y.backward(dy_used)
    ref_dx = x.grad + dres
    xg = x.detach().to(gpu)
    _, mean, rstd = ops.layernorm_fwd(xg, w.detach().to(gpu), b.detach().to(gpu), torch.float32, True)
    dgam = torch.empty(cols, device=gpu)
    dbet = torch.empty(cols, device=gpu)
    s_res, s_dx = torch.empty(cols, device=gpu), torch.empty(cols, device=gpu)
    dx, dxc = ops.layernorm_bwd(dy.to(gpu).to(cdt), xg, w.detach().to(gpu), mean, rstd, dres.to(gpu), cdt,
                                dgam, dbet, dres_sum_out=s_res, dx_sum_out=s_dx)
    assert rel_err(dx, ref_dx) < 1e-5
    assert rel_err(s_res, dres.double().sum(0)) < 1e-5 and rel_err(s_dx, ref_dx.double().sum(0)) < 1e-5
    dx_b, _ = ops.layernorm_bwd(dy.to(gpu).to(cdt), xg, w.detach().to(gpu), mean, rstd, dres.to(gpu), cdt,
                                dgam, dbet)                       # the variant without the sums
    assert torch.equal(dx_b, dx)
    assert rel_err(dgam, w.grad) < 1e-5
    assert rel_err(dbet, b.grad) < 1e-5
    if cdt in HALF:
        assert dxc is not None and rel_err(dxc.float(), ref_dx) < 2 ** -8
    else:
        assert dxc is None


# ------------------------------------------------------------------ GEMM
def ref_gemm(a, b, trans_a, b_kn):
    a = a.double()
    b = b.double()
    A = a.t() if trans_a else a
    B = b if b_kn else b.t()
    return A @ B


@pytest.fixture(params=[-1, 8, 9], ids=["auto", "p8", "p8bk64"])
def gemm_variant(request, L):
    """Run the bf16 GEMM tests three times: with the automatic kernel choice, with the persistent
    256x256 kernel forced (bf_gemm_set_variant), which the automatic choice only picks for M >= 32768, and with
    its BK = 64 full-line main loop forced (variant 9: NT calls whose K is a multiple of 64)."""
    L.lib.bf_gemm_set_variant(request.param)
    yield request.param
    L.lib.bf_gemm_set_variant(-1)


def _skip_f32_dup(cdt, gemm_variant):
    if cdt == torch.float32 and gemm_variant != -1:
        pytest.skip("the exact fp32 kernel has one variant")


GEMM_SHAPES = [  # m, n, k
    (197 * 2, 768, 768), (100, 128, 64), (275, 1536, 512), (77 * 3, 512, 2048), (130, 256, 192),
    (1000, 640, 128), (777, 384, 320)]


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_layouts(ops, gpu, m, n, k, layout, cdt, gemm_variant):
    _skip_f32_dup(cdt, gemm_variant)
    trans_a = layout[0] == "t"
    b_kn = layout[1] == "n"
    if trans_a and m % 8:          # A stored [K, M]: the leading dimension must stay 16-byte aligned
        m = (m + 7) // 8 * 8
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = torch.randn((k, m) if trans_a else (m, k), generator=g)
    b = torch.randn((k, n) if b_kn else (n, k), generator=g)
    if cdt in HALF:
        a, b = round_to(a, cdt), round_to(b, cdt)
    ref = ref_gemm(a, b, trans_a, b_kn)
    out = ops.gemm(a.to(gpu).to(cdt), b.to(gpu).to(cdt), trans_a=trans_a, b_kn=b_kn,
                   out_dtype=torch.float32, split_k=1)
    assert out.shape == (m, n)
    assert rel_err(out, ref) < 2e-6 * math.sqrt(k) + 1e-6


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("split", [1, 3, 8])
def test_gemm_splitk_tn(ops, gpu, cdt, split, gemm_variant):
    """dW = dY^T X with the token dimension (not a multiple of the K tile) split."""
    _skip_f32_dup(cdt, gemm_variant)
    g = torch.Generator().manual_seed(5)
    tokens, n_out, k_in = 275 * 3 + 1, 256, 136
    dy = torch.randn(tokens, n_out, generator=g)
    x = torch.randn(tokens, k_in, generator=g)
    if cdt in HALF:
        dy, x = round_to(dy, cdt), round_to(x, cdt)
    ref = dy.double().t() @ x.double()
    out = ops.gemm(dy.to(gpu).to(cdt), x.to(gpu).to(cdt), trans_a=True, b_kn=True, out_dtype=torch.float32,
                   split_k=split)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["none", "quick", "erf", "tanh"])
def test_gemm_epilogue_bias_act_residual(ops, L, gpu, cdt, act, gemm_variant):
    _skip_f32_dup(cdt, gemm_variant)
    g = torch.Generator().manual_seed(11)
    m, n, k = 200, 384, 128
    a, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) * k ** -0.5
    bias = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    if cdt in HALF:
        a, w = round_to(a, cdt), round_to(w, cdt)
    pre = a.double() @ w.double().t() + bias.double()
    if act == "quick":
        v = pre * torch.sigmoid(1.702 * pre)
    elif act == "erf":
        v = F.gelu(pre)
    elif act == "tanh":
        v = F.gelu(pre, approximate="tanh")
    else:
        v = pre
    ref = v + res.double()
    code = {"none": L.ACT_NONE, "quick": L.ACT_QUICK_GELU, "erf": L.ACT_ERF_GELU, "tanh": L.ACT_TANH_GELU}[act]
    got = ops.gemm(a.to(gpu).to(cdt), w.to(gpu).to(cdt), bias=bias.to(gpu), act=code,
                   want_pre=(act != "none"), residual=res.to(gpu), out_dtype=torch.float32)
    if act != "none":
        got, got_pre = got
        assert rel_err(got_pre.float(), pre) < (2e-6 if cdt == torch.float32 else 2 ** -8)
    assert rel_err(got, ref) < 5e-6
    # bf16 output + in-place residual (C aliases the residual)
    if cdt in HALF:
        o16 = ops.gemm(a.to(gpu).to(cdt), w.to(gpu).to(cdt), bias=bias.to(gpu), act=code,
                       out_dtype=cdt)
        assert rel_err(o16.float(), v) < 2 ** -8
    xres = res.to(gpu).clone()
    ops.gemm(a.to(gpu).to(cdt), w.to(gpu).to(cdt), bias=bias.to(gpu), residual=xres, out=xres)
    assert rel_err(xres, pre + res.double()) < 5e-6


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_gelu_grad_and_remap(ops, L, gpu, cdt, gemm_variant):
    _skip_f32_dup(cdt, gemm_variant)
    g = torch.Generator().manual_seed(13)
    m, n, k = 3 * 17, 256, 64
    dy, w = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g)     # dX = dY W, W stored [K,N]
    pre = torch.randn(m, n, generator=g)
    if cdt in HALF:
        dy, w, pre = round_to(dy, cdt), round_to(w, cdt), round_to(pre, cdt)
    p = pre.double().requires_grad_(True)
    F.gelu(p).backward(dy.double() @ w.double())
    got = ops.gemm(dy.to(gpu).to(cdt), w.to(gpu).to(cdt), b_kn=True, act=L.ACT_ERF_GELU_GRAD,
                   aux=pre.to(gpu).to(cdt), out_dtype=torch.float32)
    assert rel_err(got, p.grad) < 1e-5
    if cdt in HALF:      # bf16 result (the fused MLP backward): fast erf' + final rounding
        got16 = ops.gemm(dy.to(gpu).to(cdt), w.to(gpu).to(cdt), b_kn=True, act=L.ACT_ERF_GELU_GRAD,
                         aux=pre.to(gpu).to(cdt), out_dtype=cdt)
        assert rel_err(got16.float(), p.grad) < 2 ** -7
    # fused bias gradient, whatever kernel ran: a bf_colsum launch sums the STORED result; the persistent kernel's
    # GELU-grad epilogue sums the fp32 products before they are rounded for the store (round 5), i.e. what the
    # fp32-output run of the same GEMM sums to -- either way within the output's rounding of the stored columns
    o32 = None
    for odt in ((torch.float32, cdt) if cdt in HALF else (torch.float32,)):
        cs = torch.full((n,), 123.0, device=gpu)
        o = ops.gemm(dy.to(gpu).to(cdt), w.to(gpu).to(cdt), b_kn=True, act=L.ACT_ERF_GELU_GRAD,
                     aux=pre.to(gpu).to(cdt), out_dtype=odt, colsum_out=cs)
        if odt == torch.float32:
            o32 = o
            assert rel_err(cs, o.double().sum(0)) < 1e-5
        else:
            assert rel_err(cs, o.double().sum(0)) < 2 ** -7
            assert min(rel_err(cs, o.double().sum(0)), rel_err(cs, o32.double().sum(0))) < 1e-5
    # row remap: 3 batches of 17 rows scattered into [3, 20, n] at row offset 2, + broadcast table
    table = torch.randn(17, n, generator=g)
    out = torch.full((3 * 20, n), -7.0, device=gpu)
    ops.gemm(dy.to(gpu).to(cdt), w.to(gpu).to(cdt), b_kn=True, residual=table.to(gpu), resid_mod=17,
             out=out, rows_in=17, rows_out=20, row_off=2)
    ref = (dy.double() @ w.double()).reshape(3, 17, n) + table.double()
    o = out.reshape(3, 20, n).cpu()
    assert rel_err(o[:, 2:19], ref) < 1e-5
    assert (o[:, :2] == -7).all() and (o[:, 19:] == -7).all()


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("layout,k,variant", [("nt", 192, 8), ("nn", 192, 8), ("tn", 96, 8),
                                              ("nt", 64, 9), ("nt", 128, 9), ("nt", 192, 9), ("nt", 448, 9)])
def test_gemm_persistent_rounds(ops, L, gpu, layout, k, variant, half):
    """More 256x256 work items than CUs (the persistent kernel loops, the next item's first
    stages are in flight during the epilogue), ragged M / N tails, 3 K stages, bf16 and fp32 out.
    Variant 9 = the BK = 64 main loop at 1 / 2 / 3 / 7 K tiles (its region refills start and drain differently)."""
    trans_a, b_kn = layout[0] == "t", layout[1] == "n"
    m, n = 5000, 3592
    g = torch.Generator().manual_seed(77)
    a = round_to(torch.randn((k, m) if trans_a else (m, k), generator=g), half)
    b = round_to(torch.randn((k, n) if b_kn else (n, k), generator=g), half)
    bias = torch.randn(n, generator=g)
    ref = ref_gemm(a, b, trans_a, b_kn) + bias.double()
    L.lib.bf_gemm_set_variant(variant)
    try:
        for odt, tol in ((torch.float32, 1e-5), (half, 2 ** -8 if half == torch.bfloat16 else 2 ** -11)):
            out = ops.gemm(a.to(gpu).to(half), b.to(gpu).to(half), trans_a=trans_a, b_kn=b_kn,
                           bias=bias.to(gpu), out_dtype=odt, split_k=1)
            assert L.lib.bf_gemm_last_variant() == 8
            assert rel_err(out.float(), ref) < tol
    finally:
        L.lib.bf_gemm_set_variant(-1)


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("k", [1536, 1600, 2304])
def test_gemm_tail_split_inside_the_launch(ops, L, gpu, layout, k, half):
    """325 tiles on 256 CUs: the 69 tiles of the under-filled second round are cut into K chunks INSIDE the launch
    (2 chunks of 12 K tiles at K = 1536, 13 + 12 at K = 1600, 3 x 12 at K = 2304), each chunk hands its accumulators
    over through the workspace and the last to arrive sums them in chunk order and runs the epilogue.  Against the fp64 product:
    16-bit output with bias + GELU, fp32 output with bias + residual (a chunk must not preload the residual), ragged N;
    two runs bit for bit equal (fixed summation order); and the same numbers as the unsplit launch."""
    import ctypes as C
    b_kn = layout[1] == "n"
    m, n = 16640, 1096                                # 65 x 5 tiles (ragged N; the persistent kernel takes N % 8 == 0)
    g = torch.Generator().manual_seed(91 + k)
    a = round_to(torch.randn(m, k, generator=g), half)
    b = round_to(torch.randn((k, n) if b_kn else (n, k), generator=g) * k ** -0.5, half)
    bias, res = torch.randn(n, generator=g), torch.randn(m, n, generator=g)
    pre = a.double() @ (b.double() if b_kn else b.double().t()) + bias.double()
    ad, bd = a.to(gpu).to(half), b.to(gpu).to(half)
    L.lib.bf_gemm_set_variant(9)
    try:
        import ctypes as C
        o16 = ops.gemm(ad, bd, b_kn=b_kn, bias=bias.to(gpu), act=L.ACT_ERF_GELU, out_dtype=half)
        assert L.lib.bf_gemm_last_variant() == 8
        probe = L.GemmArgs()
        probe.m, probe.n, probe.k, probe.lda, probe.ldb, probe.ldc = m, n, k, k, (n if b_kn else k), n
        probe.cdt = probe.c_dtype = L.F16 if half == torch.float16 else L.BF16
        probe.b_kn, probe.split_k = int(b_kn), 1
        assert L.lib.bf_gemm_tail_workspace_bytes(C.byref(probe)) == 69 * (3 if k == 2304 else 2) * 262144
        o16b = ops.gemm(ad, bd, b_kn=b_kn, bias=bias.to(gpu), act=L.ACT_ERF_GELU, out_dtype=half)
        o32 = ops.gemm(ad, bd, b_kn=b_kn, bias=bias.to(gpu), residual=res.to(gpu), out_dtype=torch.float32)
        os.environ["BIFOLD_TEST_NO_TAIL_WS"] = "1"                 # no workspace -> the launch keeps whole tiles
        try:
            whole = ops.gemm(ad, bd, b_kn=b_kn, bias=bias.to(gpu), residual=res.to(gpu), out_dtype=torch.float32)
        finally:
            del os.environ["BIFOLD_TEST_NO_TAIL_WS"]
    finally:
        L.lib.bf_gemm_set_variant(-1)
    assert torch.equal(o16, o16b)
    assert rel_err(o16.float(), F.gelu(pre)) < (2 ** -8 if half == torch.bfloat16 else 2 ** -11)
    assert rel_err(o32, pre + res.double()) < 1e-5
    assert rel_err(o32, whole.double()) < 2e-6          # fp32 partial sums in another association, nothing else


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("shapes,k", [([(2048, 512), (512, 2048), (512, 512), (1536, 512)], 17600),      # a fusion layer's dW
                                      ([(392, 776), (768, 264)], 16416)])                               # ragged tiles
def test_grouped_weight_gradient_gemm(ops, L, gpu, half, shapes, k):
    """bf_gemm_tn_grouped: the dW = dY^T X GEMMs of one layer in one launch of the persistent kernel, against fp64 products
    of the same 16-bit operands, and bit-identical on a second run (partials are summed in split order)."""
    g = torch.Generator(device=gpu).manual_seed(k)
    probs, refs = [], []
    for m, n in shapes:
        dy = (torch.randn(k, m, device=gpu, generator=g) * 0.5).to(half)
        x = torch.randn(k, n, device=gpu, generator=g).to(half)
        out = torch.full((m, n), float("nan"), device=gpu)
        probs.append((dy, x, out))
        refs.append(dy.double().t() @ x.double())
    arr = (L.GemmTnProblem * len(probs))()
    for q, (dy, x, out) in zip(arr, probs):
        q.dy, q.x, q.c, q.m, q.n = dy.data_ptr(), x.data_ptr(), out.data_ptr(), dy.shape[1], x.shape[1]
        q.ld_dy, q.ld_x, q.ldc = dy.shape[1], x.shape[1], x.shape[1]
    assert L.lib.bf_gemm_tn_grouped_workspace_bytes(arr, len(probs), k, ops.dtype_code(probs[0][0])) > 0
    assert ops.gemm_tn_grouped(probs)
    first = [out.clone() for _, _, out in probs]
    for out, ref in zip(first, refs):
        assert rel_err(out, ref) < 2e-6 * math.sqrt(k)
    assert ops.gemm_tn_grouped(probs, alpha=0.5)
    for (_, _, out), a in zip(probs, first):
        assert torch.equal(out, a * 0.5)
    # what does not qualify is left to the caller: fp32 operands, a short contraction
    assert not ops.gemm_tn_grouped([(torch.randn(64, 16, device=gpu), torch.randn(64, 32, device=gpu),
                                     torch.empty(16, 32, device=gpu))])
    assert not ops.gemm_tn_grouped([(probs[0][0][:1024], probs[0][1][:1024], probs[0][2])])


def test_gemm_tail_split_of_last_round(ops, L, gpu):
    """270 tiles of 256x256 on 256 CUs: bf_gemm cuts the rows of the under-filled last round off and runs
    them split along K through the workspace ops.gemm passes (bf_gemm_tail_workspace_bytes); the result
    must not depend on that (bias + fp32 residual epilogue, deep K)."""
    m, n, k = 256 * 90, 768, 2048
    g = torch.Generator(device=gpu).manual_seed(91)
    a = torch.randn(m, k, device=gpu, generator=g).bfloat16()
    b = (torch.randn(n, k, device=gpu, generator=g) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=gpu, generator=g)
    res = torch.randn(m, n, device=gpu, generator=g)
    args = L.GemmArgs()
    args.m, args.n, args.k, args.lda, args.ldb, args.ldc = m, n, k, k, k, n
    args.cdt, args.c_dtype, args.split_k = L.BF16, L.F32, 1
    assert L.lib.bf_gemm_tail_workspace_bytes(args) > 0          # the plan applies to this shape
    ref = a.float() @ b.float().t() + bias + res
    out = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    assert rel_err(out, ref) < 1e-5
    out1 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, split_k=1)
    assert rel_err(out1, ref) < 1e-5


def test_colsum_and_cast(ops, gpu):
    g = torch.Generator().manual_seed(17)
    x = torch.randn(1234, 768, generator=g)
    assert rel_err(ops.colsum(x.to(gpu)), x.double().sum(0)) < 1e-5
    assert rel_err(ops.colsum(x.to(gpu).bfloat16()), bf16_round(x).double().sum(0)) < 1e-5
    acc = torch.ones(768, device=gpu)
    ops.colsum(x.to(gpu), out=acc, accumulate=True)
    assert rel_err(acc, x.double().sum(0) + 1) < 1e-5
    y = torch.randn(1001, generator=g)
    assert torch.equal(ops.cast(y.to(gpu), torch.bfloat16).cpu(), y.to(torch.bfloat16))


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("B,Lq,H,D", [(40, 275, 16, 32), (24, 197, 12, 64), (3, 700, 4, 48)])
def test_attention_fwd_many_items_lazy_rescale(ops, gpu, half, B, Lq, H, D):
    """The forward kernels at bench-like item counts, with scores spread over a wide range (x 6) so that the lazily
    moved softmax reference point is exercised: rows whose maximum keeps rising across key blocks, against fp64."""
    g = torch.Generator(device=gpu).manual_seed(B * 1000 + Lq)
    qkv = (torch.randn(B * Lq, 3 * H * D, device=gpu, generator=g) * 2.5).to(half)
    out, lse = ops.attention_fwd(qkv, B, Lq, H, D, D ** -0.5, 0, None, need_lse=True)
    q, k, v = qkv.double().reshape(B, Lq, 3, H, D).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * D ** -0.5
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * Lq, H * D)
    assert rel_err(out.float(), ref) < (2 ** -7 if half == torch.bfloat16 else 2 ** -9)
    assert rel_err(lse, torch.logsumexp(s, -1)) < 1e-5


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("top", [300.0, 1500.0])
@pytest.mark.parametrize("B,Lq,H,D", [(3, 275, 2, 32), (2, 197, 2, 64), (1, 700, 2, 48), (2, 77, 2, 64), (1, 700, 2, 64),
                                       (2, 262, 2, 48), (1, 1100, 1, 48)])
def test_attention_fwd_rows_that_keep_rising(ops, gpu, half, top, B, Lq, H, D):
    """Round 5: the forward kernels take a block's probabilities against the reference point as it stands and compute a
    row maximum only in a wave's first block or when a lane's partial row sum reaches 2^14.  Here every row's scores RISE
    with the key index, by `top` natural-log units over the sequence: 300 -> each 32-key block lifts the rows far past
    2^14 (the re-referencing path in every block); 1 500 -> past fp32's exp range (the partial sum is inf).  Against fp64."""
    g = torch.Generator().manual_seed(Lq + D)
    qkv = torch.randn(B, Lq, 3, H, D, generator=g) * 0.05
    qkv[:, :, 0] += 1.0                                                     # q ~ ones
    ramp = torch.linspace(0.0, top, Lq) / D ** 0.5                          # k_j ~ c_j * ones: s_ij ~ c_j * sqrt(D)
    qkv[:, :, 1] += ramp[None, :, None, None]
    qkv[:, :, 2] = torch.randn(B, Lq, H, D, generator=g)
    qkv = round_to(qkv.reshape(B * Lq, 3 * H * D), half)
    ref_o, ref_lse = ref_attention(qkv, B, Lq, H, D, D ** -0.5, 0, None)
    out, lse = ops.attention_fwd(qkv.to(gpu).to(half), B, Lq, H, D, D ** -0.5, 0, None, need_lse=True)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.float(), ref_o) < 1.5e-2
    assert rel_err(lse, ref_lse) < 1e-5


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("B,Lq,H,D,mode", [(2, 576, 3, 64, 0), (1, 577, 2, 64, 0), (2, 1050, 2, 48, 2), (1, 2950, 2, 48, 2),
                                            (1, 640, 1, 64, 2), (1, 705, 2, 48, 0), (3, 513, 1, 48, 0), (1, 1089, 1, 64, 0)])
def test_attention_fwd_pipelined_long_sequences(ops, gpu, half, B, Lq, H, D, mode):
    """attn_fwd_pipe_kernel (round 6, attention_pipe.hip): sequences beyond LDS at d = 48 / 64 -- workgroups of three and of
    four waves, key tails of 0 / 1 / 6 / 26 / 33 keys, waves with one live query block or none, the query-row mask with
    whole masked frames, lone masked rows and a masked first row -- output and log-sum-exp against fp64."""
    g = torch.Generator().manual_seed(B * Lq + D)
    qkv = round_to(torch.randn(B * Lq, 3 * H * D, generator=g) * 1.3, half)
    row_mask = None
    if mode == 2:
        row_mask = (torch.rand(B, Lq, generator=g) > 0.2).int()
        row_mask[:, 0] = 0
        row_mask[:, 65:65 + 300] = 0
        row_mask[0, Lq - 70:] = 0
    ref_o, ref_lse = ref_attention(qkv, B, Lq, H, D, D ** -0.5, mode, row_mask)
    out, lse = ops.attention_fwd(qkv.to(gpu).to(half), B, Lq, H, D, D ** -0.5, mode,
                                 None if row_mask is None else row_mask.to(gpu), need_lse=True)
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.float(), ref_o) < (2 ** -7 if half == torch.bfloat16 else 2 ** -9)
    assert rel_err(lse, ref_lse) < 1e-5
    if mode == 2:                                            # unmasked rows alone (the masked rows' lse is -1e5 + ln L)
        live = (row_mask == 1)[:, None, :].expand(B, H, Lq)
        assert float((lse.cpu()[live] - ref_lse[live].float()).abs().max()) < 2e-4


def ref_attention(qkv, B, Lq, H, D, scale, mode, row_mask):
    q, k, v = qkv.double().reshape(B, Lq, 3, H, D).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * scale
    if mode == 1:
        s = s.masked_fill(torch.ones(Lq, Lq, dtype=torch.bool).triu(1), float("-inf"))
    if mode == 2:
        s = s.masked_fill(row_mask[:, None, :, None] == 0, -100000.0)
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * Lq, H * D), torch.logsumexp(s, -1)


ATTN_CASES = [  # B, L, H, D, mode
    (2, 77, 2, 64, 1), (2, 197, 3, 64, 0), (2, 275, 4, 32, 0), (3, 275, 2, 32, 2), (2, 130, 2, 48, 2),
    (1, 600, 2, 48, 0), (1, 31, 1, 32, 1), (2, 64, 2, 64, 0)]


@pytest.mark.parametrize("B,Lq,H,D,mode", ATTN_CASES)
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_attention_fwd(ops, gpu, B, Lq, H, D, mode, cdt):
    g = torch.Generator().manual_seed(B * Lq + D)
    qkv = torch.randn(B * Lq, 3 * H * D, generator=g)
    if cdt in HALF:
        qkv = round_to(qkv, cdt)
    row_mask = None
    if mode == 2:
        row_mask = (torch.rand(B, Lq, generator=g) > 0.3).int()
        row_mask[:, 0] = 0
    ref_o, ref_lse = ref_attention(qkv, B, Lq, H, D, D ** -0.5, mode, row_mask)
    out, lse = ops.attention_fwd(qkv.to(gpu).to(cdt), B, Lq, H, D, D ** -0.5, mode,
                                 None if row_mask is None else row_mask.to(gpu), need_lse=True)
    # bf16: P is rounded to bf16 before P V and the output is bf16
    assert rel_err(out.float(), ref_o) < (1e-5 if cdt == torch.float32 else 1.5e-2)
    assert rel_err(lse, ref_lse) < 1e-5


@pytest.mark.parametrize("B,Lq,H,D,mode", [(2, 275, 2, 32, 0), (2, 100, 2, 32, 2), (1, 77, 2, 64, 1),
                                            (2, 90, 2, 48, 2), (1, 512, 2, 32, 0), (2, 130, 3, 32, 1),
                                            (1, 200, 2, 48, 0), (1, 300, 1, 64, 0), (1, 600, 1, 32, 2),
                                            (1, 853, 2, 48, 2), (2, 400, 2, 32, 1), (1, 1000, 1, 64, 0),
                                            (1, 353, 1, 32, 0), (1, 225, 1, 48, 1),
                                            # streamed split pair (sequences beyond LDS): causal across chunks,
                                            # ragged last chunk, every head width
                                            (1, 700, 2, 64, 1), (1, 577, 2, 48, 1), (1, 900, 1, 32, 0),
                                            (2, 576, 2, 64, 0), (1, 1290, 1, 48, 2)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_attention_bwd(ops, gpu, B, Lq, H, D, mode, cdt):
    g = torch.Generator().manual_seed(Lq + D)
    qkv = torch.randn(B * Lq, 3 * H * D, generator=g)
    dout = torch.randn(B * Lq, H * D, generator=g)
    if cdt in HALF:
        qkv, dout = round_to(qkv, cdt), round_to(dout, cdt)
    row_mask = None
    if mode == 2:
        row_mask = (torch.rand(B, Lq, generator=g) > 0.3).int()
    leaf = qkv.double().requires_grad_(True)
    ref_o, _ = ref_attention(leaf, B, Lq, H, D, D ** -0.5, mode, row_mask)
    ref_o.backward(dout.double())
    qg = qkv.to(gpu).to(cdt)
    rm = None if row_mask is None else row_mask.to(gpu)
    out, lse = ops.attention_fwd(qg, B, Lq, H, D, D ** -0.5, mode, rm, need_lse=True)
    dqkv = ops.attention_bwd(qg, out, dout.to(gpu).to(cdt), lse, B, Lq, H, D, D ** -0.5, mode, rm)
    assert rel_err(dqkv.float(), leaf.grad) < (2e-5 if cdt == torch.float32 else 2e-2)


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("B,Lq,H", [(2, 65, 2), (3, 96, 4), (2, 100, 2), (5, 197, 3), (2, 257, 2), (37, 275, 16), (2, 288, 4)])
def test_attention_bwd_one_pass_kernel_over_ragged_lengths(ops, gpu, half, B, Lq, H):
    """d = 32 without a mask, 3 - 9 key blocks: attn_bwd_fused32_kernel (waves with 1, 2 or 3 key blocks, a last
    block of 1 - 32 live keys, more workgroups than CU slots) against the fp64 formula -- and against the two-kernel
    pair's tolerance, since both must hold the same bound."""
    D = 32
    g = torch.Generator().manual_seed(Lq * 7 + H)
    qkv = round_to(torch.randn(B * Lq, 3 * H * D, generator=g) * 1.5, half)
    dout = round_to(torch.randn(B * Lq, H * D, generator=g), half)
    leaf = qkv.double().requires_grad_(True)
    ref_o, _ = ref_attention(leaf, B, Lq, H, D, D ** -0.5, 0, None)
    ref_o.backward(dout.double())
    qg = qkv.to(gpu).to(half)
    out, lse = ops.attention_fwd(qg, B, Lq, H, D, D ** -0.5, 0, None, need_lse=True)
    dqkv = ops.attention_bwd(qg, out, dout.to(gpu).to(half), lse, B, Lq, H, D, D ** -0.5, 0, None)
    got, want = dqkv.float().cpu().view(B, Lq, 3, H, D), leaf.grad.float().view(B, Lq, 3, H, D)
    for which, name in enumerate(("dQ", "dK", "dV")):
        assert rel_err(got[:, :, which], want[:, :, which]) < 2e-2, name
    again = ops.attention_bwd(qg, out, dout.to(gpu).to(half), lse, B, Lq, H, D, D ** -0.5, 0, None)
    assert torch.equal(again, dqkv)                         # partial dQ blocks are summed in wave order: bit-reproducible


FUSED_P_SCRIPT = """
import sys, torch
sys.path.insert(0, sys.argv[1])
from bifold_amd.hip import ops
gpu = torch.device("cuda:0")
D = 32
def ref(qkv, dout, B, Lq, H):
    leaf = qkv.double().requires_grad_(True)
    q, k, v = leaf.reshape(B, Lq, 3, H, D).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * Lq, H * D)
    o.backward(dout.double())
    return leaf.grad
for half in (torch.bfloat16, torch.float16):
    for B, Lq, H in ((2, 65, 2), (3, 96, 4), (5, 197, 3), (2, 257, 2), (75, 275, 16), (2, 288, 4), (130, 100, 8)):
        g = torch.Generator(device=gpu).manual_seed(Lq * 7 + H)
        qkv = (torch.randn(B * Lq, 3 * H * D, device=gpu, generator=g) * 1.5).to(half)
        dout = torch.randn(B * Lq, H * D, device=gpu, generator=g).to(half)
        want = ref(qkv, dout, B, Lq, H).float().view(B, Lq, 3, H, D)
        out, lse = ops.attention_fwd(qkv, B, Lq, H, D, D ** -0.5, 0, None, need_lse=True)
        dqkv = ops.attention_bwd(qkv, out, dout, lse, B, Lq, H, D, D ** -0.5, 0, None)
        got = dqkv.float().view(B, Lq, 3, H, D)
        for which, name in enumerate(("dQ", "dK", "dV")):
            err = ((got[:, :, which] - want[:, :, which]).abs().max() / want[:, :, which].abs().max()).item()
            assert err < 2e-2, (half, B, Lq, H, name, err)
        again = ops.attention_bwd(qkv, out, dout, lse, B, Lq, H, D, D ** -0.5, 0, None)
        assert torch.equal(again, dqkv), (half, B, Lq, H)
print("FUSED-P-OK")
"""


def test_attention_bwd_persistent_one_pass_kernel(gpu, tmp_path):
    """attn_bwd_fused32p_kernel (round 5: one twelve-wave workgroup per CU walking (batch, head) items; the default from two
    items per CU, forced here with BIFOLD_ATTN_BWD_FUSED_P=2, hence the child process): sequences of 3 - 9 key blocks (the
    nine-block ones take this kernel), a last block of 1 - 32 live keys, fewer items than CUs, and 1 200 items so that every
    workgroup walks four or five of them through the rolling refill of its images -- against the fp64 formula, twice
    (bit-reproducible)."""
    import os
    import subprocess
    import sys
    script = tmp_path / "fp.py"
    script.write_text(FUSED_P_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, str(script), root],
                       env=dict(os.environ, BIFOLD_ATTN_BWD_FUSED_P="2"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FUSED-P-OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("B,Lq,H,D,mode,spans", [
    (2, 576, 2, 64, 0, []), (1, 577, 3, 64, 0, []), (1, 1089, 1, 64, 0, []), (1, 545, 3, 48, 0, []),
    (1, 853, 2, 48, 2, []), (2, 600, 2, 48, 2, [(64, 256)]), (1, 1290, 1, 64, 2, [(0, 33), (577, 1154)]),
    (1, 2950, 2, 48, 2, [(65, 65 + 577), (65 + 2 * 577, 65 + 3 * 577)]), (1, 1050, 2, 48, 2, [(65, 65 + 3 * 197)]),
    (1, 800, 1, 48, 2, [(0, 800)])])
def test_attention_bwd_pipelined_long_sequences(ops, gpu, half, B, Lq, H, D, mode, spans):
    """attn_bwd_q_pipe_kernel + attn_bwd_kv_pipe_kernel (round 6, attention_pipe.hip): d = 48 / 64 beyond 512 tokens, no
    mask and the query-row mask -- whole masked frames (the blocks they cover entirely are not walked: their dV share is
    the summed-dO vector, their dQ a zero fill), lone masked rows inside live blocks, an odd number of live blocks, key
    and query tails, workgroups with idle waves, EVERY row masked -- against the fp64 formula, twice (bit-reproducible:
    the masked blocks' dO rows are added in a fixed order), masked queries' gradient exactly zero."""
    g = torch.Generator().manual_seed(Lq * 3 + D)
    qkv = round_to(torch.randn(B * Lq, 3 * H * D, generator=g), half)
    dout = round_to(torch.randn(B * Lq, H * D, generator=g), half)
    row_mask = None
    if mode == 2:
        row_mask = (torch.rand(B, Lq, generator=g) > 0.1).int()
        for a, b in spans:
            row_mask[:, a:b] = 0
    leaf = qkv.double().requires_grad_(True)
    ref_o, _ = ref_attention(leaf, B, Lq, H, D, D ** -0.5, mode, row_mask)
    ref_o.backward(dout.double())
    qg, dg = qkv.to(gpu).to(half), dout.to(gpu).to(half)
    rm = None if row_mask is None else row_mask.to(gpu)
    out, lse = ops.attention_fwd(qg, B, Lq, H, D, D ** -0.5, mode, rm, need_lse=True)
    dqkv = ops.attention_bwd(qg, out, dg, lse, B, Lq, H, D, D ** -0.5, mode, rm)
    got, want = dqkv.float().cpu().view(B, Lq, 3, H, D), leaf.grad.float().view(B, Lq, 3, H, D)
    for which, name in enumerate(("dQ", "dK", "dV")):
        assert rel_err(got[:, :, which], want[:, :, which]) < 2e-2, name
    if row_mask is not None:
        assert float(got[:, :, 0][row_mask == 0].abs().max()) == 0.0
    again = ops.attention_bwd(qg, out, dg, lse, B, Lq, H, D, D ** -0.5, mode, rm)
    assert torch.equal(again, dqkv)


@pytest.mark.parametrize("B,Lq,H,D,spans", [(2, 600, 2, 48, [(64, 256)]), (1, 1290, 1, 64, [(0, 33), (577, 1154)]),
                                            (2, 700, 1, 32, [(100, 420), (650, 700)])])
def test_attention_bwd_with_whole_masked_segments(ops, gpu, B, Lq, H, D, spans):
    """siglip_sequential masks whole context frames (query rows, models/siglip.py:118-130): the streamed
    backward skips 32-query blocks that are masked entirely (dQ = 0, dK untouched, dV += dO / L).  Spans
    are chosen to start and end inside blocks as well as on their boundaries."""
    g = torch.Generator().manual_seed(Lq * 3 + D)
    qkv = bf16_round(torch.randn(B * Lq, 3 * H * D, generator=g))
    dout = bf16_round(torch.randn(B * Lq, H * D, generator=g))
    row_mask = torch.ones(B, Lq, dtype=torch.int32)
    for a, b in spans:
        row_mask[:, a:b] = 0
    row_mask[0, 5] = 0                                    # and a lone masked row in a live block
    leaf = qkv.double().requires_grad_(True)
    ref_o, _ = ref_attention(leaf, B, Lq, H, D, D ** -0.5, 2, row_mask)
    ref_o.backward(dout.double())
    qg, rm = qkv.to(gpu).bfloat16(), row_mask.to(gpu)
    out, lse = ops.attention_fwd(qg, B, Lq, H, D, D ** -0.5, 2, rm, need_lse=True)
    assert rel_err(out.float(), ref_o.detach()) < 1.5e-2
    dqkv = ops.attention_bwd(qg, out, dout.to(gpu).bfloat16(), lse, B, Lq, H, D, D ** -0.5, 2, rm)
    assert rel_err(dqkv.float(), leaf.grad) < 2e-2
    dq = dqkv.float().view(B, Lq, 3, H * D)[:, :, 0]
    for a, b in spans:
        assert float(dq[:, a:b].abs().max()) == 0.0       # masked queries receive exactly no gradient


# ------------------------------------------------------------------ token plumbing
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_patchify_matches_conv(ops, gpu, cdt):
    g = torch.Generator().manual_seed(23)
    img = torch.randn(2, 3, 64, 64, generator=g)
    w = torch.randn(40, 3, 16, 16, generator=g)
    cols = ops.patchify(img.to(gpu), 16, cdt)
    ref = F.conv2d(img, w, stride=16).flatten(2).transpose(1, 2).reshape(-1, 40)
    got = cols.float().cpu() @ w.flatten(1).t()
    assert rel_err(got, ref) < (1e-5 if cdt == torch.float32 else 1e-2)
    assert torch.equal(cols.float().cpu(), (round_to(cols.float().cpu(), cdt) if cdt in HALF
                                            else cols.cpu()))


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_patchify_patch14_padded(ops, gpu, cdt):
    """ViT-L/14: 14-pixel patches, im2col rows padded with zeros from 588 to 640 columns."""
    g = torch.Generator().manual_seed(24)
    img = torch.randn(2, 3, 56, 56, generator=g)
    w = torch.randn(24, 3, 14, 14, generator=g)
    cols = ops.patchify(img.to(gpu), 14, cdt)
    assert cols.shape == (2 * 16, 640) and ops.patchify_row_length(14) == 640
    c = cols.float().cpu()
    assert (c[:, 588:] == 0).all()
    ref = F.conv2d(img, w, stride=14).flatten(2).transpose(1, 2).reshape(-1, 24)
    assert rel_err(c[:, :588] @ w.flatten(1).t(), ref) < (1e-5 if cdt == torch.float32 else 1e-2)


@pytest.mark.parametrize("cdt,tol", [(torch.float32, 3e-6), (torch.bfloat16, 1.2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("B,n,W,K", [(3, 16, 128, 192), (300, 196, 768, 768)])
def test_patch_embed_epilogue_and_lnpre_ln1(ops, gpu, observed, cdt, tol, B, n, W, K):
    """K1: the patch-embed GEMM writes patches + pos into the token stream through its row-remap epilogue, and
    bf_vision_lnpre_ln1 finishes it (class rows, ln_pre in place, ln_1 of block 0) -- against the formulas of
    clip/model.py:297-309 in fp32.  The large case dispatches the persistent GEMM (M = 58 800)."""
    g = torch.Generator().manual_seed(31)
    cols, w = torch.randn(B * n, K, generator=g), torch.randn(W, K, generator=g) * K ** -0.5
    cls, pos = torch.randn(W, generator=g), torch.randn(n + 1, W, generator=g)
    g0, b0 = 1 + 0.1 * torch.randn(W, generator=g), 0.1 * torch.randn(W, generator=g)
    g1, b1 = 1 + 0.1 * torch.randn(W, generator=g), 0.1 * torch.randn(W, generator=g)
    cc, wc = cols.to(cdt), w.to(cdt)
    patches = cc.double() @ wc.double().t()
    x_ref = torch.cat([cls.double().expand(B, 1, W), patches.reshape(B, n, W)], 1) + pos.double()
    x_ref = F.layer_norm(x_ref, (W,), g0.double(), b0.double(), 1e-5)
    h_ref = F.layer_norm(x_ref, (W,), g1.double(), b1.double(), 1e-5)
    posg = pos.to(gpu)
    x = torch.full((B * (n + 1), W), float("nan"), dtype=torch.float32, device=gpu)   # every row must be written
    ops.gemm(cc.to(gpu), wc.to(gpu), residual=posg[1:], resid_mod=n, out=x, rows_in=n, rows_out=n + 1, row_off=1)
    h = ops.vision_lnpre_ln1(x, cls.to(gpu), posg[0], g0.to(gpu), b0.to(gpu), g1.to(gpu), b1.to(gpu), cdt, B, n)
    ex = rel_err(x.reshape(B, n + 1, W), x_ref.float())
    eh = rel_err(h.float().reshape(B, n + 1, W), h_ref.float())
    observed(f"k1_patch_embed_lnpre_ln1_{str(cdt).split('.')[-1]}_B{B}", x_rel=ex, h_rel=eh)
    assert ex < 3e-6 and eh < tol
    # and the same values as the unfused path (separate GEMM, bf_vision_assemble, LayerNorm)
    p2 = ops.gemm(cc.to(gpu), wc.to(gpu), out_dtype=torch.float32)
    x2 = ops.vision_assemble(p2, cls.to(gpu), posg, g0.to(gpu), b0.to(gpu), B, n)
    assert rel_err(x, x2) < 2e-6


def test_vision_assemble_embed_fusion(ops, gpu):
    g = torch.Generator().manual_seed(29)
    B, n, W = 3, 16, 128
    patches, cls, pos = torch.randn(B * n, W, generator=g), torch.randn(W, generator=g), torch.randn(n + 1, W, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(W, generator=g), 0.1 * torch.randn(W, generator=g)
    x = torch.cat([cls.expand(B, 1, W), patches.reshape(B, n, W)], 1) + pos
    ref = F.layer_norm(x, (W,), gam, bet, 1e-5)
    got = ops.vision_assemble(patches.to(gpu), cls.to(gpu), pos.to(gpu), gam.to(gpu), bet.to(gpu), B, n)
    assert rel_err(got.reshape(B, n + 1, W), ref) < 2e-6
    # embedding
    table, tpos = torch.randn(50, 64, generator=g), torch.randn(7, 64, generator=g)
    tok = torch.randint(0, 50, (B, 7), generator=g).int()
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    e = ops.embed_tokens(tok.to(gpu), table.to(gpu), tpos.to(gpu), flag)
    assert torch.equal(e.reshape(B, 7, 64).cpu(), table[tok.long()] + tpos)
    assert flag.item() == 0
    bad = tok.clone(); bad[0, 0] = 50
    ops.embed_tokens(bad.to(gpu), table.to(gpu), tpos.to(gpu), flag)
    assert flag.item() == 1
    # fusion assemble + backward
    nt, nr, D = 5, 9, 64
    tf, rf = torch.randn(B * nt, D, generator=g), torch.randn(B * nr, D, generator=g)
    tt, tp = torch.randn(D, generator=g), torch.randn(nt + 1, D, generator=g)
    rp, ty = torch.randn(nr, D, generator=g), torch.randn(2, D, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (tf, rf, tt, tp, rp, ty)]
    tf_, rf_, tt_, tp_, rp_, ty_ = leaves
    xt = torch.cat([tt_.expand(B, 1, D), tf_.reshape(B, nt, D)], 1) + tp_ + ty_[0]
    xr = rf_.reshape(B, nr, D) + rp_ + ty_[1]
    xref = torch.cat([xt, xr], 1)
    got = ops.fusion_assemble(*(t.to(gpu) for t in (tf, rf, tt, tp, rp, ty)), B, nt, nr)
    assert rel_err(got.reshape(B, 1 + nt + nr, D), xref) < 1e-6
    dx = torch.randn(B, 1 + nt + nr, D, generator=g)
    xref.backward(dx)
    d_tt, d_tp = torch.empty(D, device=gpu), torch.empty(nt + 1, D, device=gpu)
    d_rp, d_ty = torch.empty(nr, D, device=gpu), torch.empty(2, D, device=gpu)
    d_rgb = ops.fusion_assemble_bwd(dx.to(gpu), torch.float32, d_tt, d_tp, d_rp, d_ty, B, nt, nr)
    assert rel_err(d_rgb, rf_.grad) < 1e-6
    assert rel_err(d_tt, tt_.grad) < 1e-5 and rel_err(d_tp, tp_.grad) < 1e-5
    assert rel_err(d_rp, rp_.grad) < 1e-5 and rel_err(d_ty, ty_.grad) < 1e-5


def test_segment_assemble_and_backward(ops, gpu):
    g = torch.Generator().manual_seed(41)
    B, T, n, D = 3, 2, 5, 64
    txt = torch.randn(B, 4, D, generator=g)
    ctx = torch.randn(B, T, n, D, generator=g)
    img = torch.randn(B, n, D, generator=g)
    leaves = [t.requires_grad_(True) for t in (torch.randn(D, generator=g), torch.randn(D, generator=g),
                                                torch.randn(T * (n + 1), D, generator=g), torch.randn(2, D, generator=g))]
    tt, it, cpos, ty = leaves
    xt = torch.cat([tt.expand(B, 1, D), txt], 1) + ty[0]
    xc = torch.cat([it.expand(B, T, 1, D), ctx], 2).reshape(B, T * (n + 1), D) + cpos + ty[1]
    xi = torch.cat([it.expand(B, 1, D), img], 1) + ty[1]
    ref = torch.cat([xt, xc, xi], 1)
    seq = ref.shape[1]
    x = torch.empty(B * seq, D, device=gpu)
    ctx_g = ctx.to(gpu)
    ttg, itg, cposg, tyg = (t.detach().to(gpu) for t in leaves)
    ops.segment_assemble(txt.to(gpu), 4 * D, ttg, None, tyg[0], x, B, 4, seq, 0)
    off = 5
    for f in range(T):
        ops.segment_assemble(ctx_g[:, f], T * n * D, itg, cposg[f * (n + 1):(f + 1) * (n + 1)], tyg[1], x, B, n, seq, off)
        off += n + 1
    ops.segment_assemble(img.to(gpu), n * D, itg, None, tyg[1], x, B, n, seq, off)
    assert rel_err(x.view(B, seq, D), ref) < 1e-6
    dx = torch.randn(B, seq, D, generator=g)
    ref.backward(dx)
    d_tt, d_it = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
    d_cpos, d_ty = torch.empty(T * (n + 1), D, device=gpu), torch.zeros(2, D, device=gpu)
    dxg = dx.to(gpu).reshape(B * seq, D)
    ops.segment_assemble_bwd(dxg, d_tt, None, d_ty[0], B, 4, seq, 0)
    off = 5
    for f in range(T):
        ops.segment_assemble_bwd(dxg, d_it, d_cpos[f * (n + 1):(f + 1) * (n + 1)], d_ty[1], B, n, seq, off)
        off += n + 1
    ops.segment_assemble_bwd(dxg, d_it, None, d_ty[1], B, n, seq, off)
    assert rel_err(d_tt, tt.grad) < 1e-5 and rel_err(d_it, it.grad) < 1e-5
    assert rel_err(d_cpos, cpos.grad) < 1e-5 and rel_err(d_ty, ty.grad) < 1e-5


# ------------------------------------------------------------------ decoder + BCE
@pytest.mark.parametrize("dim,grid,heads", [(512, 14, 2), (96, 3, 4), (768, 24, 1), (640, 2, 4), (512, 1, 2)])
def test_decoder_collapsed_matches_as_written(ops, gpu, dim, grid, heads):
    from oracle import bifold_oracle as O
    from bifold.models.decoder_math import upsample_operator
    g = torch.Generator().manual_seed(dim + grid)
    B = 2
    state = {k: v for k, v in O.make_tail_state(g, dim=dim, depth=0, is_bimanual=(heads == 4)).items()
             if "decoder" in k}
    names = O.decoder_heads(heads == 4)[:heads]
    tok = torch.randn(B, 7 + grid * grid + 1, dim, generator=g)       # slice [7:] is the image modality
    img = tok[:, 7:]
    feat = img[:, 1:].reshape(B, grid, grid, dim).permute(0, 3, 1, 2)
    chans = [dim, dim // 2, dim // 2, dim // 4, dim // 4, 1]
    U = upsample_operator(grid).to(gpu)
    assert rel_err(U, O.upsample_operator(grid)) < 1e-7
    w_eff = torch.empty(heads, dim, device=gpu)
    b_eff = torch.empty(heads, device=gpu)
    rows_ws = torch.empty(heads, 4 * dim, device=gpu)
    params = []
    leaves = {}
    for h, nme in enumerate(names):
        ws = [state[f"pick_place.{nme}_decoder.decoder_net.{i}.weight"].reshape(chans[j + 1], chans[j])
              for j, i in enumerate((0, 2, 4, 6, 8))]
        bs = [state[f"pick_place.{nme}_decoder.decoder_net.{i}.bias"] for i in (0, 2, 4, 6, 8)]
        params.append(([w.to(gpu).contiguous() for w in ws], [b.to(gpu) for b in bs]))
        ops.decoder_collapse_fwd(params[-1][0], params[-1][1], w_eff[h], b_eff[h:h + 1], rows_ws[h])
    tok_g = tok.to(gpu)
    L_all = tok.shape[1]
    img_ptr_view = tok_g[:, 7:]
    low = ops.decoder_project(img_ptr_view, L_all * dim, w_eff, b_eff, B, heads, grid * grid, dim)
    heat = ops.decoder_upsample_sigmoid(low, U, heads * B, grid).reshape(heads, B, grid * 16, grid * 16)
    # reference: the decoder as written, with autograd for the backward check
    leaf_tok = tok.clone().requires_grad_(True)
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    feat = leaf_tok[:, 8:].reshape(B, grid, grid, dim).permute(0, 3, 1, 2)
    refs = [torch.sigmoid(O.conv_decoder_logits(st, f"pick_place.{n}_decoder.", feat).squeeze(1)) for n in names]
    for h in range(heads):
        assert rel_err(heat[h], refs[h]) < 2e-5
    d_heat = torch.randn(heads, B, grid * 16, grid * 16, generator=g) * 1e-3
    torch.autograd.backward(refs, [d_heat[h] for h in range(heads)])
    g_low = ops.decoder_upsample_sigmoid_bwd(heat, d_heat.to(gpu), U, heads * B, grid)
    d_tok = torch.full((B, L_all, dim), 5.0, device=gpu)
    g_w, g_b = torch.empty(heads, dim, device=gpu), torch.empty(heads, device=gpu)
    ops.decoder_project_bwd(img_ptr_view, L_all * dim, w_eff, g_low, d_tok[:, 7:], L_all * dim, g_w, g_b, B,
                            heads, grid * grid, dim)
    assert rel_err(d_tok[:, 7:], leaf_tok.grad[:, 7:]) < 5e-5
    assert (d_tok[:, :7] == 5.0).all()
    for h, nme in enumerate(names):
        dws = [torch.empty_like(w) for w in params[h][0]]
        dbs = [torch.empty_like(b) for b in params[h][1]]
        ops.decoder_collapse_bwd(params[h][0], params[h][1], rows_ws[h], g_w[h], g_b[h:h + 1], dws, dbs)
        for j, i in enumerate((0, 2, 4, 6, 8)):
            rw = st[f"pick_place.{nme}_decoder.decoder_net.{i}.weight"].grad.flatten(1)
            rb = st[f"pick_place.{nme}_decoder.decoder_net.{i}.bias"].grad
            assert rel_err(dws[j], rw) < 1e-4, (nme, i)
            assert rel_err(dbs[j], rb) < 1e-4, (nme, i)


def test_bce_forward_backward(ops, gpu):
    g = torch.Generator().manual_seed(31)
    p = torch.rand(5, 64, 64, generator=g)
    t = torch.rand(5, 64, 64, generator=g)
    p[0, 0, :4] = torch.tensor([0.0, 1.0, 1e-30, 1.0 - 1e-7])       # clamp edges
    t[0, 0, :4] = torch.tensor([1.0, 0.0, 1.0, 0.0])
    leaf = p.clone().requires_grad_(True)
    ref = torch.nn.BCELoss()(leaf, t)
    (ref * 0.7).backward()
    loss = ops.bce_mean(p.to(gpu), t.to(gpu))
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    d = ops.bce_mean_bwd(p.to(gpu), t.to(gpu), torch.tensor([0.7], device=gpu))
    assert rel_err(d, leaf.grad) < 1e-6


# ------------------------------------------------------------------ optimiser
@pytest.mark.parametrize("half", HALF)
@pytest.mark.parametrize("decoupled,wd", [(False, 0.0), (True, 0.01), (False, 0.01)])
def test_adam_matches_torch(ops, gpu, decoupled, wd, half):
    g = torch.Generator().manual_seed(37)
    n = 4096 + 4
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref_p], lr=1e-3, weight_decay=wd)
    p, m, v = p0.to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    shadow = torch.empty(n, dtype=half, device=gpu)
    for step in range(1, 5):
        gr = torch.randn(n, generator=g)
        ref_p.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.to(gpu), m, v, shadow, step, 1e-3, 0.9, 0.999, 1e-8, wd, decoupled)
    assert rel_err(p, ref_p.detach()) < 1e-6
    assert torch.equal(shadow.cpu(), p.cpu().to(half))
    ss = ops.sumsq(p)
    assert abs(ss.item() - (p.double() ** 2).sum().item()) < 1e-4 * ss.item()
    c = ops.clip_coef(ss, 1.0)
    assert abs(c.item() - min(1.0, 1.0 / (math.sqrt(ss.item()) + 1e-6))) < 1e-6


def test_scale_check_scales_in_place_and_counts_nonfinite(ops, gpu):
    """bf_scale_check: the two ends of the fp16 mode's loss scaling (x *= S exactly for a power of two; inf / nan counted)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4099, generator=g)
    xd = x.to(gpu)
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    ops.scale_check_(xd[:4096], 2.0 ** 20, flag)
    assert torch.equal(xd[:4096].cpu(), x[:4096] * 2.0 ** 20) and torch.equal(xd[4096:].cpu(), x[4096:]) and flag.item() == 0
    ops.scale_check_(xd[:4096], 2.0 ** -20, flag)
    assert torch.equal(xd.cpu(), x) and flag.item() == 0
    xd[7], xd[100], xd[4000] = float("inf"), float("nan"), 3e38
    ops.scale_check_(xd[:4096], 4.0, flag)                      # 3e38 * 4 overflows to inf as well
    assert flag.item() == 3
    y = torch.randn(6, device=gpu)                               # count % 4 != 0: the tail elements are scaled too
    y0 = y.clone()
    ops.scale_check_(y, 0.5)
    assert torch.equal(y, y0 * 0.5)


def test_guarded_adam_step_drops_flagged_steps_and_keeps_its_own_count(ops, gpu):
    """The fp16 mode's guarded step (round 4): bf_adam_step with a non-zero device flag changes NOTHING (parameters,
    moments, 16-bit copy); bf_gradscale_update then halves the device-side scale, counts the drop and clears the
    flag; later steps take the drop off Adam's t on the device -- the result equals torch.optim.Adam fed only the
    clean gradients, although the host kept counting 1, 2, 3 ... and never read anything back."""
    g = torch.Generator().manual_seed(21)
    n = 4096
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p, m, v = p0.to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    shadow = torch.zeros(n, dtype=torch.float16, device=gpu)
    state = torch.zeros(8, device=gpu)
    state[0], state[1] = 2.0 ** 20, 2.0 ** -20
    flag = torch.zeros(1, dtype=torch.int32, device=gpu)
    sched = dict(backoff=0.5, growth=2.0, growth_interval=2, min_scale=1.0, max_scale=2.0 ** 20)
    drops = {2, 3, 6}
    for call in range(1, 9):
        gr = torch.randn(n, generator=g)
        before = (p.clone(), m.clone(), v.clone(), shadow.clone())
        if call in drops:
            gd = gr.to(gpu)
            gd[5] = float("nan")
            flag.fill_(3)                          # what bf_scale_check would have counted
        else:
            gd = gr.to(gpu)
            ref_p.grad = gr.clone()
            opt.step()
        ops.adam_step(p, gd, m, v, shadow, call, 1e-3, 0.9, 0.999, 1e-8, 0.0, False, None, flag, state[3:4], 0.0)
        ops.gradscale_update_(state, flag, **sched)
        if call in drops:
            for a, b in zip(before, (p, m, v, shadow)):
                assert torch.equal(a, b)
        assert flag.item() == 0
    assert torch.isfinite(p).all()
    assert rel_err(p, ref_p.detach()) < 1e-6
    st = state.tolist()
    # 2^20 -> (drop) 2^19 -> (drop) 2^18 -> two clean steps: 2^19 -> (drop) 2^18 -> two clean steps: 2^19
    assert st[3] == 3 and st[4] == 9 and st[0] == 2.0 ** 19 and st[1] == 2.0 ** -19 and st[5] == 0
    # the scale read from the device: x *= state[0]
    x = torch.ones(8, device=gpu)
    ops.scale_check_(x, state[0:1], flag)
    assert torch.equal(x.cpu(), torch.full((8,), 2.0 ** 19)) and flag.item() == 0
    # at the floor an overflow is recorded as such
    state[0] = 1.0
    flag.fill_(1)
    ops.gradscale_update_(state, flag, **sched)
    assert state[0].item() == 1.0 and state[5].item() == 1


# ------------------------------------------------------------------ either side of the step
def _postproc():
    import os

    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "postproc.npz"))


def test_heatmap_pick_matches_reference_fixture(ops):
    """bf_heatmap_pick against the reference's sample_from_heatmap outputs: exact (index work)."""
    import numpy as np
    z = _postproc()
    mask = torch.from_numpy(z["mask"]).cuda()
    xy, conf = ops.heatmap_pick(torch.from_numpy(z["uni.pick_heatmap"]).cuda(), mask)
    assert np.array_equal(xy.cpu().numpy(), z["uni.expect.pick"])
    assert np.array_equal(conf.cpu().numpy(), z["uni.expect.pick_conf"])
    xy, _ = ops.heatmap_pick(torch.from_numpy(z["uni.place_heatmap"]).cuda())
    assert np.array_equal(xy.cpu().numpy(), z["uni.expect.place"])


@pytest.mark.parametrize("size,batch", [(224, 5), (64, 3), (7, 2)])
def test_heatmap_pick_matches_oracle(ops, size, batch):
    import numpy as np
    from oracle import bifold_oracle as O
    g = torch.Generator().manual_seed(size)
    heat = torch.rand(batch, size, size, generator=g)
    heat[0] = (heat[0] * 4).floor() / 4                 # heavy ties
    heat[1, size // 2, size // 3] = float("nan")        # numpy.argmax picks the first NaN
    mask = (torch.rand(batch, 1, size, size, generator=g) > 0.97).float()
    mask[:, 0, size - 1, size - 1] = 1.0
    for m in (None, mask):
        xy, conf = ops.heatmap_pick(heat.cuda(), None if m is None else m.cuda())
        exy, econf = O.sample_from_heatmap(heat.numpy(), None if m is None else m.numpy())
        assert np.array_equal(xy.cpu().numpy(), exy)
        assert np.array_equal(conf.cpu().numpy(), econf, equal_nan=True)
    # an empty mask is an error in the reference; the kernel flags it instead of faulting
    xy, _ = ops.heatmap_pick(heat.cuda(), torch.zeros_like(mask).cuda())
    assert (xy.cpu() == -2).all()


def test_gaussian_targets_match_reference_fixture(ops):
    """bf_gaussian_targets against Processor._make_gaussmap outputs.  fp64 exp on the device may
    differ from numpy's in the last fp64 ulp, which can flip the final fp32 rounding: tolerance
    1 fp32 ulp (1.2e-7 relative), stated here."""
    import numpy as np
    z = _postproc()
    for strategy in ("first", "gmm", "fit"):
        names = [k.split(".")[2] for k in z.files if k.startswith(f"gauss.{strategy}.") and k.endswith(".points")]
        assert names, strategy
        pmax = max(z[f"gauss.{strategy}.{n}.points"].shape[0] for n in names)
        pts = torch.zeros(len(names) + 1, pmax, 2)
        cnt = torch.zeros(len(names) + 1, dtype=torch.int32)
        for i, n in enumerate(names):
            p = torch.from_numpy(z[f"gauss.{strategy}.{n}.points"])
            pts[i, :p.shape[0]] = p
            cnt[i] = p.shape[0]
        maps = ops.gaussian_targets(pts.cuda(), cnt.cuda(), 40, 5, strategy).cpu().numpy()
        for i, n in enumerate(names):
            exp = z[f"gauss.{strategy}.{n}.expect"]
            assert np.allclose(maps[i], exp, rtol=2.4e-7, atol=1e-37), (strategy, n, np.abs(maps[i] - exp).max())
        assert not maps[-1].any()                        # count 0: the absent-arm map


def test_gaussian_targets_full_size_properties(ops):
    """224x224, B*heads maps in one launch: peak exactly 1 at the rounded pixel, symmetric."""
    from oracle import bifold_oracle as O
    g = torch.Generator().manual_seed(5)
    pts = torch.rand(64, 3, 2, generator=g) * 223
    cnt = torch.randint(1, 4, (64,), generator=g, dtype=torch.int32)
    for strategy in ("first", "gmm"):
        maps = ops.gaussian_targets(pts.cuda(), cnt.cuda(), 224, 5.0, strategy).cpu()
        assert maps.max().item() == 1.0 and maps.min().item() >= 0.0
        for i in (0, 17, 63):
            exp = torch.from_numpy(O.make_gaussmap(pts[i, :cnt[i]].numpy(), 224, 5.0, strategy))
            assert torch.allclose(maps[i], exp, rtol=2.4e-7, atol=1e-37)
    # fit: three points per map (non-singular covariance), the unnormalised density of the fitted normal
    cnt3 = torch.full((64,), 3, dtype=torch.int32)
    maps = ops.gaussian_targets(pts.cuda(), cnt3.cuda(), 224, 5.0, "fit").cpu()
    for i in (0, 17, 63):
        exp = torch.from_numpy(O.make_gaussmap(pts[i].numpy(), 224, 5.0, "fit"))
        assert torch.allclose(maps[i], exp, rtol=4.8e-7, atol=1e-37)
    with pytest.raises(ValueError):
        ops.gaussian_targets(pts.cuda(), cnt.cuda(), 224, 5.0, "bogus")


def test_preprocess_rgb_matches_oracle(ops, gpu):
    """Mask fill + scale + normalise (the tensor half of the reference's Processor) against the numpy
    restatement: fp32, tolerance 2 ulp (1 / std is applied as a multiplication)."""
    import numpy as np
    from oracle import bifold_oracle as O
    rng = np.random.default_rng(7)
    raw = rng.integers(0, 256, (3, 48, 48, 3), dtype=np.uint8)
    mask = (rng.random((3, 48, 48)) > 0.4).astype(np.float32)
    mean, std = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    for mk in (None, mask):
        got = ops.preprocess_rgb(torch.from_numpy(raw).to(gpu), None if mk is None else torch.from_numpy(mk).to(gpu),
                                 mean, std).cpu().numpy()
        for i in range(3):
            exp = O.preprocess_rgb(raw[i], None if mk is None else mk[i], mean, std)
            assert np.allclose(got[i], exp, rtol=3e-7, atol=3e-7)


# ------------------------------------------------------------------ fp8 (OCP e4m3) forward path of the frozen towers
def _e4m3_bytes(t):
    return t.to(torch.float8_e4m3fn).view(torch.uint8)


def test_fp8_gemm_is_exact_on_integer_operands(ops, gpu):
    """Layout check of v_mfma_scale_f32_16x16x128_f8f6f4 (guide: test MFMA maps with exact, ASYMMETRIC data): small
    integers are exact in e4m3, so the result must equal the integer matrix product bit for bit, incl. M / N tails."""
    g = torch.Generator().manual_seed(3)
    m, n, k = 200, 136, 384
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    b = torch.randint(-3, 4, (n, k), generator=g).float()
    b[5] += torch.arange(k) % 3                       # rows / columns that differ systematically (asymmetry)
    a[:, 7] = 4.0
    ref = a.double() @ b.double().t()
    one_m, one_n = torch.ones(m, device=gpu), torch.ones(n, device=gpu)
    c = ops.gemm_fp8(_e4m3_bytes(a).to(gpu), one_m, _e4m3_bytes(b).to(gpu), one_n, out_dtype=torch.float32)
    assert torch.equal(c.double().cpu(), ref)
    # scales and bias are applied per row / per column
    sa = (torch.rand(m, generator=g) + 0.5).to(gpu)
    sb = (torch.rand(n, generator=g) + 0.5).to(gpu)
    bias = torch.randn(n, generator=g).to(gpu)
    c2 = ops.gemm_fp8(_e4m3_bytes(a).to(gpu), sa, _e4m3_bytes(b).to(gpu), sb, bias=bias, out_dtype=torch.float32)
    want = ref * sa.double().cpu()[:, None] * sb.double().cpu()[None, :] + bias.double().cpu()
    assert rel_err(c2, want) < 1e-6


@pytest.mark.parametrize("k", [128, 384, 768, 1024])
def test_fp8_persistent_gemm_is_exact_on_integer_operands(ops, L, gpu, k):
    """The persistent 256 x 256 e4m3 kernel (M >= 16 384 and at least one tile per two CUs): integer operands are exact in
    e4m3 and in the fp32 accumulator, the bf16 output holds integers up to 256 -- so every element must equal the integer
    product bit for bit: fragment layout across the stage PAIR a K = 128 MFMA reads, ring hand-over between pairs (1, 3, 6,
    8 pairs), ticket order, ragged M / N tails.  Then scales, bias and activation against fp64."""
    g = torch.Generator().manual_seed(11 + k)
    m, n = 256 * 70 - 56, 1024 - 24                    # 70 x 4 tiles, ragged last row / column tile
    a = torch.randint(-1, 2, (m, k), generator=g).float()
    b = torch.randint(-1, 2, (n, k), generator=g).float()
    a[:, 5] = 0.0                                      # keep |sums| small: bf16 holds integers up to 256 exactly
    a[3::7, 9] = 2.0
    b[2::5, 9] = 2.0                                   # asymmetric structure along m, n and k
    ref = a.double() @ b.double().t()
    assert ref.abs().max() < 256
    one_m, one_n = torch.ones(m, device=gpu), torch.ones(n, device=gpu)
    a8, b8 = _e4m3_bytes(a).to(gpu), _e4m3_bytes(b).to(gpu)
    c = ops.gemm_fp8(a8, one_m, b8, one_n)
    assert c.dtype == torch.bfloat16 and torch.equal(c.double().cpu(), ref)
    L.lib.bf_gemm_set_variant(-1)
    sa = (torch.rand(m, generator=g) + 0.5).to(gpu)
    sb = (torch.rand(n, generator=g) + 0.5).to(gpu)
    bias = torch.randn(n, generator=g).to(gpu)
    c2 = ops.gemm_fp8(a8, sa, b8, sb, bias=bias, act=L.ACT_QUICK_GELU)
    z = ref * sa.double().cpu()[:, None] * sb.double().cpu()[None, :] + bias.double().cpu()
    assert rel_err(c2.float(), z * torch.sigmoid(1.702 * z)) < 2 ** -7


def test_fp8_quantisers_match_torch_e4m3(ops, gpu):
    g = torch.Generator().manual_seed(4)
    w = torch.randn(300, 768, generator=g) * torch.logspace(-2, 1, 300).view(-1, 1)        # rows of very different size
    q, s = ops.quantize_rows_fp8(w.to(gpu))
    want_s = w.abs().amax(1) / 448.0
    assert rel_err(s, want_s) < 1e-6
    want_q = _e4m3_bytes(w / want_s[:, None])
    assert (q.cpu() != want_q).float().mean().item() < 1e-3          # ties / last-ulp of the division only
    deq = q.cpu().view(torch.float8_e4m3fn).float() * s.cpu()[:, None]
    assert ((deq - w).abs() <= 0.0626 * w.abs() + want_s[:, None] * 2 ** -9).all()   # 3 mantissa bits (+ subnormals)
    # LayerNorm straight to e4m3
    x = torch.randn(77 * 3, 768, generator=g) * 2 + 0.3
    gam, bet = 1 + 0.2 * torch.randn(768, generator=g), 0.2 * torch.randn(768, generator=g)
    y8, ys = ops.layernorm_fwd_fp8(x.to(gpu), gam.to(gpu), bet.to(gpu))
    ref = F.layer_norm(x, (768,), gam, bet, 1e-5)
    assert rel_err(ys, ref.abs().amax(1) / 448.0) < 1e-5
    deq = y8.cpu().view(torch.float8_e4m3fn).float() * ys.cpu()[:, None]
    assert ((deq - ref).abs() <= 0.0626 * ref.abs() + ys.cpu()[:, None] * 2 ** -9 + 1e-5).all()


@pytest.mark.parametrize("m,n,k,act", [(50432 // 16, 2304, 768, 0), (1000, 3072, 768, 1), (333, 1536, 512, 0),
                                       (197 * 96, 2304, 768, 0), (77 * 256, 2048, 512, 1)])      # the last two: persistent kernel
def test_fp8_gemm_matches_dequantised_product(ops, L, gpu, m, n, k, act):
    """Against fp64 on the DEQUANTISED operands (what is left is accumulation order and the bf16 output rounding)."""
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.05
    bias = torch.randn(n, generator=g) * 0.1
    a8, sa = ops.quantize_rows_fp8(x.to(gpu))
    b8, sb = ops.quantize_rows_fp8(w.to(gpu))
    c = ops.gemm_fp8(a8, sa, b8, sb, bias=bias.to(gpu), act=L.ACT_QUICK_GELU if act else L.ACT_NONE)
    da = a8.cpu().view(torch.float8_e4m3fn).double() * sa.double().cpu()[:, None]
    db = b8.cpu().view(torch.float8_e4m3fn).double() * sb.double().cpu()[:, None]
    ref = da @ db.t() + bias.double()
    if act:
        ref = ref * torch.sigmoid(1.702 * ref)
    assert c.dtype == torch.bfloat16 and rel_err(c.float(), ref) < 2 ** -7
    # and against the unquantised product: the price of e4m3 operands with row / channel scales
    full = x.double() @ w.double().t() + bias.double()
    if act:
        full = full * torch.sigmoid(1.702 * full)
    assert rel_err(c.float(), full) < 5e-2


@pytest.mark.parametrize("m,n,k", [(300, 256, 128), (16384 + 200, 512, 256)])     # 128x128 kernels / persistent kernel
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_gelu_savegrad_and_mul_aux(ops, L, gpu, m, n, k, cdt):
    """BF_ACT_ERF_GELU_SAVEGRAD: C = gelu(v) and pre = gelu'(v) from one erf evaluation; BF_ACT_MUL_AUX: v *= aux --
    the pair that replaces (ERF_GELU + saved pre, ERF_GELU_GRAD) in the fusion MLP (modules.py:21-27 and its autograd)."""
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.2
    b = torch.randn(n, generator=g) * 0.3
    xc, wc = (x, w) if cdt == torch.float32 else (round_to(x, cdt), round_to(w, cdt))
    v = (xc.double() @ wc.double().t() + b.double()).requires_grad_(True)
    ref = F.gelu(v)
    ref.sum().backward()
    a, gp = ops.gemm(x.to(gpu).to(cdt), w.to(gpu).to(cdt), bias=b.to(gpu), act=L.ACT_ERF_GELU_SAVEGRAD, want_pre=True)
    tol = 1e-5 if cdt == torch.float32 else 2 ** -7
    assert rel_err(a.float(), ref) < tol and rel_err(gp.float(), v.grad) < tol
    # backward half: dX-like product times the saved derivative, against the same product times gelu'(pre) in fp64
    dy = torch.randn(m, k, generator=g)
    w2 = torch.randn(k, n, generator=g) * 0.2                  # stored [K, N]: b_kn
    dyc, w2c = (dy, w2) if cdt == torch.float32 else (round_to(dy, cdt), round_to(w2, cdt))
    want = (dyc.double() @ w2c.double()) * gp.double().cpu()
    got = ops.gemm(dy.to(gpu).to(cdt), w2.to(gpu).to(cdt), b_kn=True, act=L.ACT_MUL_AUX, aux=gp)
    assert rel_err(got.float(), want) < tol
    if cdt in HALF and m >= 16384:                   # the fused bias-gradient sums ride along on this path
        cs = torch.empty(n, dtype=torch.float32, device=gpu)
        got2 = ops.gemm(dy.to(gpu).to(cdt), w2.to(gpu).to(cdt), b_kn=True, act=L.ACT_MUL_AUX, aux=gp, colsum_out=cs)
        # the epilogue sums the fp32 products BEFORE they are rounded for the store (round 5): the exact column sums to
        # fp32 accumulation, and the stored columns' sums to the output's rounding (bf16: 16 584 roundings of 2^-9 each)
        assert torch.equal(got2, got)
        assert rel_err(cs, want.sum(0)) < 1e-4
        assert rel_err(cs, got2.float().double().sum(0)) < 2 ** -7


@pytest.mark.parametrize("rows,cols", [(197 * 3 + 1, 768), (77 * 2, 512), (9, 1024), (300, 256)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_add_layernorm_fwd(ops, gpu, rows, cols, out_dtype):
    """x_out = x + delta (bf16 branch output), y = LN(x_out): the residual add of clip/model.py:217-218 fused into
    the LayerNorm that follows it; in place and out of place, with statistics."""
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.5
    delta = (torch.randn(rows, cols, generator=g) * 0.7).to(out_dtype if out_dtype in HALF else torch.bfloat16)
    w = 1 + 0.2 * torch.randn(cols, generator=g)
    b = 0.2 * torch.randn(cols, generator=g)
    xs = x.double() + delta.double()
    ref = F.layer_norm(xs, (cols,), w.double(), b.double(), 1e-5)
    xd = x.to(gpu)
    y, mean, rstd, xo = ops.add_layernorm_fwd(xd, delta.to(gpu), w.to(gpu), b.to(gpu), out_dtype, save_stats=True)
    assert xo is xd and rel_err(xd, xs) < 1e-6                      # updated in place
    assert rel_err(y.float(), ref) < (2e-6 if out_dtype == torch.float32 else 2 ** -8)
    assert rel_err(mean, xs.mean(1)) < 1e-5
    x2 = x.to(gpu)
    out = torch.empty_like(x2)
    y2, _, _, xo2 = ops.add_layernorm_fwd(x2, delta.to(gpu), w.to(gpu), b.to(gpu), out_dtype, x_out=out)
    assert xo2 is out and torch.equal(x2.cpu(), x) and torch.equal(out, xd) and torch.equal(y2, y)


@pytest.mark.parametrize("rows,cols", [(197 * 3 + 1, 768), (77 * 2, 512), (9, 1024)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_add_layernorm_fwd_lazy_stream_is_the_two_pass_form(ops, gpu, rows, cols, out_dtype):
    """The frozen towers write their residual stream once per block (bf_add2_layernorm_fwd): a pass that only produces
    LN(x + a) and a pass that adds both branches, (x + a) + m.  Same BITS as the two read-modify-write passes
    (clip/model.py:215-219: x = x + attn; x = x + mlp)."""
    g = torch.Generator().manual_seed(7 * rows + cols)
    ddt = out_dtype if out_dtype in HALF else torch.bfloat16
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.5).to(gpu)
    a = (torch.randn(rows, cols, generator=g) * 0.7).to(ddt).to(gpu)
    m = (torch.randn(rows, cols, generator=g) * 0.9).to(ddt).to(gpu)
    w2, b2 = (1 + 0.2 * torch.randn(cols, generator=g)).to(gpu), (0.2 * torch.randn(cols, generator=g)).to(gpu)
    w1, b1 = (1 + 0.2 * torch.randn(cols, generator=g)).to(gpu), (0.2 * torch.randn(cols, generator=g)).to(gpu)
    # two passes, the stream updated in place each time
    xt = x.clone()
    h2_want, _, _, _ = ops.add_layernorm_fwd(xt, a, w2, b2, out_dtype)
    h1_want, mean_want, rstd_want, _ = ops.add_layernorm_fwd(xt, m, w1, b1, out_dtype, save_stats=True)
    # lazy: the first pass leaves the stream alone, the second adds both branches
    xl = x.clone()
    h2, _, _, xo = ops.add_layernorm_fwd(xl, a, w2, b2, out_dtype, write_x=False)
    assert xo is None and torch.equal(xl, x) and torch.equal(h2, h2_want)
    h1, mean, rstd, xo = ops.add_layernorm_fwd(xl, a, w1, b1, out_dtype, delta2=m, save_stats=True)
    assert xo is xl and torch.equal(xl, xt) and torch.equal(h1, h1_want)
    assert torch.equal(mean, mean_want) and torch.equal(rstd, rstd_want)
    # and without a stream at all (the tower's last block)
    xn = x.clone()
    y, _, _, xo = ops.add_layernorm_fwd(xn, a, w1, b1, out_dtype, delta2=m, write_x=False)
    assert xo is None and torch.equal(xn, x) and torch.equal(y, h1_want)


# ---- replayable dropout masks and the LoRA adapter-branch kernels (csrc/lora.hip, ABI 11) --------------------------------
@pytest.mark.parametrize("cdt", [torch.float32, torch.float16, torch.bfloat16])
def test_dropout_applies_the_mask_bf_dropout_mask_replays(ops, gpu, cdt):
    """bf_dropout / bf_dropout_mask share one keep function (common.h: one hash per four elements, 15 bits each): the
    output is x / (1 - p) exactly where the replayed mask says 1, the kept fraction is 1 - p within sampling error, the
    four elements of a hash group are not tied to each other, and another seed gives another mask."""
    torch.manual_seed(3)
    n, p = 1 << 20, 0.25
    x = (torch.randn(n, device=gpu) + 3.0).to(cdt)
    y = ops.dropout(x, p, 1234)
    m = ops.dropout_mask((n,), p, 1234, gpu).bool()
    want = torch.where(m, (x.float() / (1.0 - p)), torch.zeros((), device=gpu)).to(cdt)
    assert torch.equal(y, want)
    frac = m.float().mean().item()
    assert abs(frac - (1.0 - p)) < 4.0 * math.sqrt(p * (1.0 - p) / n)
    g = m.view(-1, 4).float()
    for a in range(4):
        for b in range(a + 1, 4):                         # independent fields of one hash: joint keep rate = (1 - p)^2
            joint = (g[:, a] * g[:, b]).mean().item()
            assert abs(joint - (1.0 - p) ** 2) < 5e-3
    m2 = ops.dropout_mask((n,), p, 1235, gpu).bool()
    assert 0.3 < (m ^ m2).float().mean().item() < 0.45     # 2 p (1 - p) = 0.375
    acc = torch.ones_like(x)
    ops.dropout(x, p, 1234, out=acc, accumulate=True)
    assert torch.equal(acc, (1.0 + torch.where(m, x.float() / (1.0 - p), torch.zeros((), device=gpu))).to(cdt))   # one rounding


def _lora_case(gpu, cdt, rows, width, n_slots, rank, ld_extra):
    g = torch.Generator().manual_seed(rows * 7 + width + n_slots)
    ld = n_slots * width + ld_extra
    wide = torch.randn(rows, ld, generator=g).to(cdt).to(gpu)
    wt = (torch.randn(n_slots * rank, width, generator=g) / math.sqrt(width)).to(cdt).to(gpu)
    u = torch.randn(rows, n_slots * rank, generator=g).to(gpu)
    return wide, wt, u


@pytest.mark.parametrize("cdt", HALF)
@pytest.mark.parametrize("rows,width,n_slots,rank", [(7, 64, 2, 8), (1000, 768, 2, 8), (333, 256, 3, 8), (130, 512, 1, 16),
                                                     (257, 128, 2, 16), (4096, 768, 1, 8)])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_lora_kernels_against_the_masked_matrix_products(ops, gpu, cdt, rows, width, n_slots, rank, p):
    """bf_lora_down / bf_lora_up / bf_lora_tn against float64 products of the same 16-bit operands, masks replayed with
    bf_dropout_mask; both addressing forms (all slots on the same columns: the forward's h1 and the backward's dh1; one
    column block per slot: the packed projection output and its gradient), ragged row counts, a row stride beyond the
    columns used.  fp32 accumulation: 2e-5 relative (down / tn); up rounds once to the storage type."""
    wide, wt, u = _lora_case(gpu, cdt, rows, width, n_slots, rank, 8)
    seeds = [11 + 5 * s for s in range(n_slots)]
    keep = 1.0 / (1.0 - p)
    masks = [ops.dropout_mask((rows, width), p, sd, gpu).double() if p > 0 else torch.ones(rows, width, device=gpu).double()
             for sd in seeds]
    W = wt.double()
    for offs in ([0] * n_slots, [s * width for s in range(n_slots)]):
        xs = [wide[:, o:o + width].double() * m for o, m in zip(offs, masks)]
        # down
        got, got16 = ops.lora_down(wide, offs, wt, rank, width, p, seeds, alpha=keep, want_half=True)
        want = torch.cat([keep * x @ W[s * rank:(s + 1) * rank].t() for s, x in enumerate(xs)], dim=1)
        assert rel_err(got, want) < 2e-5, ("down", offs)
        assert torch.equal(got16, got.to(cdt)) and torch.equal(got, ops.lora_down(wide, offs, wt, rank, width, p, seeds, alpha=keep))
        # tn, both output layouts
        for transposed in (False, True):
            outs = [torch.full((width, rank) if transposed else (rank, width), float("nan"), device=gpu) for _ in xs]
            ops.lora_tn(wide, offs, u, outs, rank, width, transposed, p, seeds, alpha=0.5)
            for s, x in enumerate(xs):
                ref = 0.5 * u[:, s * rank:(s + 1) * rank].double().t() @ x
                assert rel_err(outs[s], ref.t() if transposed else ref) < 2e-5, ("tn", offs, transposed, s)
            again = [torch.empty_like(o) for o in outs]
            ops.lora_tn(wide, offs, u, again, rank, width, transposed, p, seeds, alpha=0.5)
            assert all(torch.equal(a, b) for a, b in zip(outs, again))          # fixed summation order
        # up: in place, nothing outside the slots' columns touched
        y = wide.clone()
        ops.lora_up(u, wt, y, offs, rank, width, p, seeds, alpha=keep)
        ref = wide.double().clone()
        for s, (o, m) in enumerate(zip(offs, masks)):
            ref[:, o:o + width] += keep * m * (u[:, s * rank:(s + 1) * rank].double() @ W[s * rank:(s + 1) * rank])
        assert torch.equal(y.float(), ref.to(cdt).float()) or rel_err(y, ref) < (2 ** -7 if cdt == torch.bfloat16 else 2 ** -10)
        if len(set(offs)) == 1 and n_slots * rank <= 16:
            # the backward's fused form: lora_tn(x, u) and lora_up(u, wt, y) in one walk; x and y are different tensors
            x2 = wide[:, :width].contiguous()
            y2 = wide[:, 8:8 + width].contiguous()
            y2_before = y2.clone()
            outs = [torch.full((rank, width), float("nan"), device=gpu) for _ in range(n_slots)]
            ops.lora_bwd_input(x2, u, wt, y2, outs, rank, width, p, seeds, alpha=keep)
            ref_y = y2_before.double()
            for sl_, m in enumerate(masks):
                us = u[:, sl_ * rank:(sl_ + 1) * rank].double()
                assert rel_err(outs[sl_], keep * us.t() @ (x2.double() * m)) < 2e-5, ("bwd_input dA", sl_)
                ref_y += keep * m * (us @ W[sl_ * rank:(sl_ + 1) * rank])
            assert torch.equal(y2.float(), ref_y.to(cdt).float()) or rel_err(y2, ref_y) < (2 ** -7 if cdt == torch.bfloat16 else 2 ** -10)
        untouched = torch.ones(wide.shape[1], dtype=torch.bool)
        for o in set(offs):
            untouched[o:o + width] = False
        assert torch.equal(y[:, untouched.to(gpu)], wide[:, untouched.to(gpu)])


def test_lora_kernels_refuse_what_they_do_not_take(ops, gpu):
    wide, wt, u = _lora_case(gpu, torch.float16, 64, 64, 2, 8, 0)
    with pytest.raises(RuntimeError):                     # fp32 storage: the GEMM form serves it
        ops.lora_down(wide.float(), [0, 0], wt.float(), 8, 64)
    with pytest.raises(RuntimeError):                     # rank 4
        ops.lora_down(wide, [0, 0, 0], wt[:12].contiguous(), 4, 64)
    with pytest.raises(RuntimeError):                     # overlapping, unequal column blocks
        ops.lora_up(u, wt, wide.clone(), [0, 32], 8, 64)
    assert not ops.lora_supported(torch.float32, 768, 8, 2) and ops.lora_supported(torch.float16, 768, 8, 2)
    assert not ops.lora_supported(torch.float16, 768, 4, 2) and not ops.lora_supported(torch.float16, 776, 8, 2)


@pytest.mark.parametrize("cdt", HALF)
@pytest.mark.parametrize("m,n,k,r,act", [(16384, 2304, 768, 16, 0), (20000, 1536, 512, 32, 0), (16640, 768, 768, 16, 4)])
def test_gemm_low_rank_addend_rides_in_the_persistent_kernel(ops, gpu, cdt, m, n, k, r, act):
    """bf_gemm_args::lr_u / lr_w (ABI 11): c = act(a b^T + lr_u lr_w^T + bias) with the low-rank product as one more MFMA
    K-step in front of the epilogue of the persistent kernel (the LoRA up-projection inside the QKV GEMM).  Against the
    float64 product of the same 16-bit operands; rows that are not a multiple of the 256-row tile; the call is refused
    where another kernel would run it."""
    from bifold_amd.hip import lib as L
    g = torch.Generator().manual_seed(m + n)
    a = (torch.randn(m, k, generator=g) * 0.5).to(cdt).to(gpu)
    b = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(cdt).to(gpu)
    u = torch.randn(m, r, generator=g).to(cdt).to(gpu)
    wl = (torch.randn(n, r, generator=g) * 0.3).to(cdt).to(gpu)
    wl[n // 3: 2 * n // 3] = 0                               # the k_proj rows of a packed projection carry no adapter
    bias = torch.randn(n, generator=g).to(gpu)
    assert ops.gemm_lowrank_ok(m, n, k, cdt, r, act)
    got = ops.gemm(a, b, bias=bias, act=act, lr_u=u, lr_w=wl)
    assert L.lib.bf_gemm_last_variant() == 8
    plain = ops.gemm(a, b, bias=bias, act=act)
    ref = a.double() @ b.double().t() + u.double() @ wl.double().t() + bias.double()
    ref0 = a.double() @ b.double().t() + bias.double()
    if act == 4:                                             # BF_ACT_TANH_GELU
        ref, ref0 = F.gelu(ref, approximate="tanh"), F.gelu(ref0, approximate="tanh")
    tol = 2 ** -7 if cdt == torch.bfloat16 else 2 ** -10
    assert rel_err(got, ref) < tol and rel_err(plain, ref0) < tol
    assert rel_err(got, ref0) > 20 * tol                     # the addend is not small here: it was really added
    # refused where the call runs on another kernel: few rows, fp32 output, a residual
    assert not ops.gemm_lowrank_ok(4096, n, k, cdt, r) and not ops.gemm_lowrank_ok(m, n, k, cdt, 8)
    with pytest.raises(RuntimeError):
        ops.gemm(a[:4096], b, bias=bias, lr_u=u[:4096].contiguous(), lr_w=wl)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b, bias=bias, lr_u=u, lr_w=wl, out_dtype=torch.float32)


# ---- the mask head's product and the mask losses (csrc/decoder.hip, ABI 12) ---------------------------------------------
@pytest.mark.parametrize("heads,detach,with_dmask,skip_head", [(1, False, True, False), (2, False, False, False),
                                                               (2, True, True, False), (2, False, True, True)])
def test_mask_mul_fwd_bwd(ops, gpu, heads, detach, with_dmask, skip_head):
    """out[h] = s[h] * m and its backward against autograd (models/pickplace.py:76-98,107-119: the pick heatmaps times
    the mask heatmap, detached or not; a head the loss ignores hands None)."""
    g = torch.Generator().manual_seed(heads * 7 + detach)
    B, S = 3, 32
    s = torch.rand(heads, B, S, S, generator=g).requires_grad_(True)
    m = torch.rand(B, S, S, generator=g).requires_grad_(True)
    out = s * (m.detach() if detach else m)
    got = ops.mask_mul_fwd(s.detach().to(gpu), m.detach().to(gpu))
    assert torch.equal(got.cpu(), out.detach())
    d_out = torch.randn(heads, B, S, S, generator=g)
    d_mask_in = torch.randn(B, S, S, generator=g) if with_dmask else None
    if skip_head:
        d_out[1] = 0.0
    total = (out * d_out).sum() + ((m * d_mask_in).sum() if with_dmask else 0.0)
    total.backward()
    d_list = [None if (skip_head and h == 1) else d_out[h].to(gpu) for h in range(heads)]
    d_s, d_m = ops.mask_mul_bwd(d_list, s.detach().to(gpu), m.detach().to(gpu),
                                d_mask_in.to(gpu) if with_dmask else None, detach)
    assert rel_err(d_s, s.grad) < 1e-6
    ref_dm = m.grad if m.grad is not None else torch.zeros_like(m)
    assert (d_m.cpu() - ref_dm).abs().max().item() <= 1e-6 * max(1.0, ref_dm.abs().max().item())
    with pytest.raises(RuntimeError, match="heads"):
        ops.mask_mul_bwd([None] * (heads + 1), s.detach().to(gpu), m.detach().to(gpu), None, detach)


@pytest.mark.parametrize("dst_dtype", [torch.float32, torch.float16])
def test_rows_accumulate(ops, gpu, dst_dtype):
    g = torch.Generator().manual_seed(5)
    B, seq, n, D = 3, 11, 5, 64
    src = torch.randn(B, n, D, generator=g)
    dst = torch.randn(B, seq, D, generator=g).to(dst_dtype)
    ref = dst.float().clone()
    ref[:, seq - n:] += src
    d = dst.to(gpu)
    ops.rows_accumulate(src.to(gpu), d[:, seq - n:])
    assert torch.equal(d[:, :seq - n].cpu(), dst[:, :seq - n])
    assert rel_err(d.float(), ref.to(dst_dtype).float()) < (1e-7 if dst_dtype == torch.float32 else 1e-3)
    with pytest.raises(RuntimeError, match="contiguous"):       # rows two elements apart
        ops.rows_accumulate(src.to(gpu), d.as_strided((B, n, D), (seq * D, 2 * D, 1)))


@pytest.mark.parametrize("B,S", [(3, 64), (5, 224)])
def test_dice_and_focal_losses(ops, gpu, B, S):
    """bf_dice_* / bf_focal_* against the reference's formulas under autograd (losses/__init__.py:126-155), with the
    probabilities at the edges torch's BCE clamps (p = 0, p = 1), an empty mask, and other alpha / gamma than the defaults."""
    from oracle import bifold_oracle as O
    g = torch.Generator().manual_seed(B * S)
    p = torch.rand(B, S, S, generator=g)
    p[0, 0, :4] = torch.tensor([0.0, 1.0, 1e-30, 1.0 - 1e-7])
    t = (torch.rand(B, 1, S, S, generator=g) > 0.6).float()
    t[1] = 0.0
    pl = p.clone().requires_grad_(True)
    ref, _ = O.dice_loss({"mask_heatmap": pl}, {"mask": t})
    ref.backward()
    loss, stats = ops.dice_fwd(p.flatten(1).to(gpu), t.flatten(1).to(gpu))
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    gs = torch.full((1,), 0.5, device=gpu)
    d = ops.dice_bwd(p.flatten(1).to(gpu), t.flatten(1).to(gpu), stats, gs)
    assert rel_err(d.view(B, S, S), 0.5 * pl.grad) < 1e-5
    for alpha, gamma in ((0.25, 2.0), (-1.0, 2.0), (0.6, 1.5), (0.25, 0.0), (0.25, 1.0)):
        pl = p.clone().requires_grad_(True)
        ref, _ = O.focal_loss({"mask_heatmap": pl}, {"mask": t}, alpha=alpha, gamma=gamma)
        ref.backward()
        got = ops.focal_fwd(p.to(gpu), t.squeeze(1).to(gpu), alpha, gamma)
        assert abs(got.item() - ref.item()) < 2e-5 * abs(ref.item()), (alpha, gamma)
        d = ops.focal_bwd(p.to(gpu), t.squeeze(1).contiguous().to(gpu), gs, alpha, gamma)
        want = 0.5 * pl.grad
        finite = torch.isfinite(want)
        # (at p = 0 with t = 1 torch's own chain 0 * inf is nan for gamma < 1; everywhere else the values agree)
        assert rel_err(torch.where(finite, d.cpu(), torch.zeros_like(want)), torch.where(finite, want, torch.zeros_like(want))) < 2e-5, (alpha, gamma)

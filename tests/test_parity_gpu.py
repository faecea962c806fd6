"""Parity of the HIP path at the things the bench actually runs, and of the reference-held per-op
fixtures THROUGH the HIP kernels (round-1 verdict: they only ever reached the CPU oracle).

 * config A at B = 256 in bf16 -- the exact bench configuration, i.e. the shapes that dispatch the
   persistent 256x256 GEMM, the two-stream towers, the GELU'+colsum epilogue at M = 70 400: heatmaps
   of 8 of the 256 samples against the CPU oracle (forward samples are independent), the loss and
   every gradient against the fp32-exact HIP mode (itself pinned to the reference at B = 2);
 * tests/golden/per_op.npz (LayerNorm 512 / 768, both GELUs, a causal CLIP block, the decoder as
   written, BCE clamp edges, Adam) and fusion_masked_bimanual.npz (the reference's own
   PickPlaceConvDecoder call with a query-row mask) through the C ABI.

Tolerances are ~3x the errors observed on MI355X (recorded by the `observed` fixture).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def load_fixture(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    groups = {}
    for k in z.files:
        head, _, rest = k.partition("/")
        groups.setdefault(head, {})[rest] = z[k]
    return groups


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def config_a(dtype):
    return dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
                patch_size=16, depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
                pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype=dtype)


def test_config_a_batch_256_bf16_matches_oracle_and_exact_mode(gpu, observed):
    """The bench configuration itself (BASELINE configs[1]: B = 256) in all three compute modes: fp16 (what bench.py
    times since round 3), bf16 and the exact fp32 mode."""
    from oracle import bifold_oracle as O
    from bifold.losses import Losses
    from bifold.models import Models
    B, N_CHECK = 256, 8
    gen = torch.Generator().manual_seed(3407)
    state = {**O.make_clip_state(gen), **O.make_tail_state(gen)}
    sample = O.synthetic_sample(B, torch.Generator().manual_seed(3418))
    dev_sample = {k: v.to(gpu) for k, v in sample.items()}
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
    got = {}
    for dtype in ("bf16", "fp16", "fp32"):
        with pytest.warns(UserWarning):
            model = Models.get_by_name(config_a(dtype), device=gpu)
        missing, unexpected = model.load_state_dict(state, strict=False)
        assert not missing and not unexpected
        model = model.to(gpu).train()
        out = model(dev_sample)
        loss, _ = loss_fn(out, dev_sample)
        loss.backward()
        torch.cuda.synchronize()
        got[dtype] = dict(loss=loss.item(),
                          heat={k: out[k][:N_CHECK].detach().float().cpu() for k in ("pick_heatmap", "place_heatmap")},
                          grad={n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()
                                if p.requires_grad})
        del model, out, loss
        torch.cuda.empty_cache()
    # ---- oracle (fp32 CPU, the reference's arithmetic) on 8 of the 256 samples
    sub = {k: v[:N_CHECK] for k, v in sample.items()}
    with torch.no_grad():
        ref = O.rgb_clip_forward(state, sub, heads=16)
    # ---- the same restatement under torch autograd on ALL 256 samples, evaluated by torch's own fp32 kernels on the
    # GPU (rocBLAS / ATen: nothing of libbifold_hip): loss and every trainable gradient at the bench's shapes, i.e. what
    # the persistent dgrad / wgrad GEMMs at M = 70 400 produce is compared with something outside the library
    trainable = sorted(got["fp32"]["grad"])
    gstate = {k: (v.to(gpu).requires_grad_(True) if k in trainable else v.to(gpu)) for k, v in state.items()}
    # in chunks of 32 samples (the loss is a mean, so chunk c contributes n_c / B of its own mean loss and gradient):
    # at B = 256 the as-written decoder's [B, 128, 224, 224] intermediates are 6.6 GB per head, and torch's own
    # kernels were seen to return a different loss and gradients there while agreeing to 1e-6 at B = 8
    CH = 32
    ggrads, gloss, torch_heat = {k: torch.zeros_like(state[k]) for k in trainable}, 0.0, 0.0
    for c0 in range(0, B, CH):
        csample = {k: v[c0:c0 + CH] for k, v in dev_sample.items()}
        gout = O.rgb_clip_forward(gstate, csample, heads=16)
        closs, _ = O.bce_gaussmap(gout, csample, False)
        closs = closs * (csample["rgb"].shape[0] / B)
        for k, g in zip(trainable, torch.autograd.grad(closs, [gstate[k] for k in trainable])):
            ggrads[k] += g.float().cpu()
        gloss += closs.item()
        if c0 == 0:
            torch_heat = max(rel(gout[k][:N_CHECK], ref[k]) for k in ("pick_heatmap", "place_heatmap"))
        del gout, closs
    del gstate
    torch.cuda.empty_cache()
    assert torch_heat < 1e-4                        # torch-on-GPU fp32 == torch-on-CPU fp32 up to summation order

    def logit(t):
        return torch.logit(t.double().clamp(1e-300, 1 - 1e-16))
    err, lerr = {}, {}
    for dtype in ("bf16", "fp16", "fp32"):
        err[dtype] = max(rel(got[dtype]["heat"][k], ref[k]) for k in ("pick_heatmap", "place_heatmap"))
        lerr[dtype] = max(rel(logit(got[dtype]["heat"][k]), logit(ref[k])) for k in ("pick_heatmap", "place_heatmap"))
    mse = max(((got["bf16"]["heat"][k].double() - ref[k].double()) ** 2).mean().item()
              for k in ("pick_heatmap", "place_heatmap"))
    loss_rel = {d: abs(got[d]["loss"] - gloss) / abs(gloss) for d in got}
    grad_err = {d: {n: rel(got[d]["grad"][n], g) for n, g in ggrads.items()} for d in got}
    worst = {d: max(grad_err[d], key=grad_err[d].get) for d in got}
    observed("config_a_b256", heat_rel_bf16=err["bf16"], heat_rel_fp32=err["fp32"], heat_mse_bf16=mse,
             logit_rel_bf16=lerr["bf16"], logit_rel_fp32=lerr["fp32"],
             loss_rel_bf16=loss_rel["bf16"], loss_rel_fp32=loss_rel["fp32"],
             worst_grad_rel_bf16=grad_err["bf16"][worst["bf16"]], worst_grad_bf16=worst["bf16"],
             worst_grad_rel_fp32=grad_err["fp32"][worst["fp32"]], worst_grad_fp32=worst["fp32"],
             median_grad_rel_bf16=float(np.median(list(grad_err["bf16"].values()))),
             heat_rel_fp16=err["fp16"], logit_rel_fp16=lerr["fp16"], loss_rel_fp16=loss_rel["fp16"],
             worst_grad_rel_fp16=grad_err["fp16"][worst["fp16"]], worst_grad_fp16=worst["fp16"],
             median_grad_rel_fp16=float(np.median(list(grad_err["fp16"].values()))))
    # the exact mode meets north_star's gate at the bench shapes on heatmaps, LOGITS, loss and every gradient
    assert err["fp32"] < 1e-3 and lerr["fp32"] < 1e-3 and loss_rel["fp32"] < 1e-4
    assert grad_err["fp32"][worst["fp32"]] < 1e-3, (worst["fp32"], grad_err["fp32"][worst["fp32"]])
    # bf16, observed on MI355X: heatmaps 2.4e-4, loss 2.2e-5, worst gradient tensor 6.8e-3 (median 2.8e-3);
    # LOGITS at random initialisation 2.6e-3 .. 3.1e-3 of max |logit| = 0.17 -- above north_star's 1e-3: the bound
    # asserted here is ~3x the observed value so that it cannot drift silently (the fp16 mode is the one that meets it)
    assert err["bf16"] < 1e-3                       # the bf16 mode meets north_star's 1e-3 gate on HEATMAPS at the bench shapes
    assert lerr["bf16"] < 8e-3, lerr["bf16"]
    assert loss_rel["bf16"] < 1e-4
    assert grad_err["bf16"][worst["bf16"]] < 2e-2, (worst["bf16"], grad_err["bf16"][worst["bf16"]])
    # fp16 (CLIP's native dtype; 2^-11 operand rounding, loss-scaled backward): north_star's 1e-3 on heatmaps AND logits
    assert err["fp16"] < 1e-3 and lerr["fp16"] < 1e-3, (err["fp16"], lerr["fp16"])
    assert loss_rel["fp16"] < 1e-4
    assert grad_err["fp16"][worst["fp16"]] < 6e-3, (worst["fp16"], grad_err["fp16"][worst["fp16"]])


def test_config_a_batch_256_fp16_logits_after_training(gpu, observed):
    """The bench configuration in the dtype bench.py times, on weights that have MOVED: ten Adam steps on the B = 256 batch
    (what the timed region does), then the first 4 samples against the CPU oracle handed the trained state_dict.  bench.py
    prints this as `parity_trained`; here it is asserted (verdict, round 5: 6d).  Logits of a trained model are larger than
    at random initialisation (max |logit| 0.17 there), which is what "relative" divides by -- both ends must hold."""
    from oracle import bifold_oracle as O
    from bifold.losses import Losses
    from bifold.models import Models
    from bifold.optim import Optimizers
    B, N_CHECK, STEPS = 256, 4, 10
    gen = torch.Generator().manual_seed(3407)
    state = {**O.make_clip_state(gen), **O.make_tail_state(gen)}
    sample = O.synthetic_sample(B, torch.Generator().manual_seed(3418))
    dev_sample = {k: v.to(gpu) for k, v in sample.items()}
    with pytest.warns(UserWarning):
        model = Models.get_by_name(config_a("fp16"), device=gpu)
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not missing and not unexpected
    model = model.to(gpu).train()
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
    opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                                 params=[p for p in model.parameters() if p.requires_grad])
    first = None
    for _ in range(STEPS):
        out = model(dev_sample)
        loss, _ = loss_fn(out, dev_sample)
        first = loss.item() if first is None else first
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert loss.item() < first and model._grad_scaler.read()["skipped"] == 0      # it trained, and no step was dropped
    model.eval()
    with torch.no_grad():
        out = model(dev_sample)
    trained = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    sub = {k: v[:N_CHECK] for k, v in sample.items()}
    with torch.no_grad():
        ref = O.rgb_clip_forward(trained, sub, heads=16)

    def logit(t):
        return torch.logit(t.double().clamp(1e-300, 1 - 1e-16))
    heat = max(rel(out[k][:N_CHECK].float().cpu(), ref[k]) for k in ("pick_heatmap", "place_heatmap"))
    lg = max(rel(logit(out[k][:N_CHECK].float().cpu()), logit(ref[k])) for k in ("pick_heatmap", "place_heatmap"))
    observed("config_a_b256_trained", heat_rel_fp16=heat, logit_rel_fp16=lg, steps=STEPS)
    assert heat < 1e-3 and lg < 1e-3, (heat, lg)


def test_reference_per_op_fixtures_through_the_hip_kernels(gpu, golden_dir, observed):
    from bifold_amd.hip import lib as L
    from bifold_amd.hip import ops
    from bifold.losses import bce_mean
    from bifold.models import Components
    from bifold.models.clip_encoder import CLIPEncoder
    from bifold.optim import Optimizers
    fx = load_fixture(golden_dir, "per_op.npz")
    seen = {}
    # LayerNorm 512 / 768 (clip/model.py:178-184, modules.py:11)
    for w in (512, 768):
        g = {k: torch.from_numpy(v).to(gpu) for k, v in fx[f"ln{w}"].items()}
        y, _, _ = ops.layernorm_fwd(g["x"], g["w"], g["b"], torch.float32)
        seen[f"ln{w}"] = rel(y, g["y"])
        assert seen[f"ln{w}"] < 1e-5
    # QuickGELU / erf-GELU live in the GEMM epilogue: x in column 0 of A times a one-hot B reproduces x exactly
    x = torch.from_numpy(fx["act"]["x"])
    a = torch.zeros(x.numel(), 64)
    a[:, 0] = x
    b = torch.zeros(64, 64)
    b[:, 0] = 1.0
    for act, key in ((L.ACT_QUICK_GELU, "quick_gelu"), (L.ACT_ERF_GELU, "erf_gelu")):
        y = ops.gemm(a.to(gpu), b.to(gpu), act=act)
        seen[key] = rel(y[:, 5], fx["act"][key])
        assert seen[key] < 1e-5, key
    # one causal ResidualAttentionBlock (clip/model.py:192-219): width 128, 2 heads, 77 tokens
    cb = fx["clipblock"]
    enc = CLIPEncoder(embed_dim=32, image_resolution=32, vision_layers=1, vision_width=64, vision_patch_size=16,
                      context_length=77, vocab_size=64, transformer_width=128, transformer_heads=2,
                      transformer_layers=1)
    sd = enc.state_dict()
    for k, v in cb.items():
        if k.startswith("state/b."):
            sd["transformer.resblocks.0." + k[len("state/b."):]] = torch.from_numpy(v)
    enc.load_state_dict(sd)
    enc = enc.to(gpu)
    for cdt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 3e-2)):
        enc._runtime(cdt)
        xin = torch.from_numpy(cb["x"]).to(gpu).view(-1, 128).clone()
        y = enc._tower(xin, "transformer.resblocks", 1, 2, 2, 77, cdt, causal=True)
        seen[f"clipblock_{'f32' if cdt == torch.float32 else 'bf16'}"] = rel(y.view(2, 77, 128), cb["y"])
        assert rel(y.view(2, 77, 128), cb["y"]) < tol
    # ConvDecoder AS WRITTEN by the reference (five 1x1 convs, four bilinear x2) vs the collapsed HIP decoder
    d = fx["decoder"]
    pp = Components.get_by_name("pick_place_convdecoder", dim=512, is_bimanual=False, fusion_model="concat_transformer",
                                num_patches=196, patch_size=16, heads=16, depth=1, dropout=0.0, mlp_ratio=4)
    sd = pp.state_dict()
    for k, v in d.items():
        if k.startswith("state/d."):
            sd["pick_decoder." + k[len("state/d."):]] = torch.from_numpy(v)
    pp.load_state_dict(sd)
    pp = pp.to(gpu)
    from bifold.models.pickplace import decoder_heads_fwd
    feat = torch.from_numpy(d["feat"])                                   # [1, 512, 14, 14]
    tok = torch.zeros(1, 197, 512)
    tok[:, 1:] = feat.flatten(2).transpose(1, 2)                         # row-major patch tokens, token 0 dropped
    tok = tok.to(gpu)
    heat, _, _ = decoder_heads_fwd(pp, tok, 197 * 512, 1)
    want = torch.sigmoid(torch.from_numpy(d["logits"]).double()).squeeze(1)
    seen["decoder"] = rel(heat[0], want)
    assert seen["decoder"] < 1e-4
    # nn.BCELoss incl. the -100 clamp at p = 0 / p = 1 (losses/__init__.py:61)
    bce = fx["bce"]
    loss = bce_mean(torch.from_numpy(bce["p"]).view(1, -1).to(gpu), torch.from_numpy(bce["t"]).view(1, -1).to(gpu))
    seen["bce"] = abs(loss.item() - float(bce["loss"])) / abs(float(bce["loss"]))
    assert seen["bce"] < 1e-6
    # four Adam steps (optim/__init__.py:7-12 -> torch.optim.Adam)
    ad = fx["adam"]
    p = torch.nn.Parameter(torch.from_numpy(ad["p0"]).to(gpu))
    opt = Optimizers.get_by_name(dict(name="adam", lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=0), [p])
    for i in range(4):
        opt.arena.grad_view(p).copy_(torch.from_numpy(ad["grads"][i]).to(gpu))
        opt.step()
    seen["adam"] = rel(p.detach(), ad["p4"])
    assert seen["adam"] < 1e-6
    observed("per_op_through_hip", **seen)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 1.5e-2)])
def test_masked_three_input_component_call_matches_reference(gpu, golden_dir, observed, dtype, tol):
    """The reference's own `PickPlaceConvDecoder(x_text, x_ctx, x_img, attention_masks=m, modalities=[0, 1, 1])`
    call (models/pickplace.py:59-124 -> fusion.py:35-64 with the query-row mask of transformer.py:33-35),
    through the STANDALONE forward of the components -- Components.get_by_name("pick_place_convdecoder") and
    ("concat_transformer") are usable on their own, as in the reference."""
    from bifold.models import Components
    fx = load_fixture(golden_dir, "fusion_masked_bimanual.npz")
    heads = int(fx["meta"]["heads"])
    pp = Components.get_by_name("pick_place_convdecoder", dim=96, is_bimanual=True, fusion_model="concat_transformer",
                                num_patches=9, patch_size=16, heads=heads, depth=2, dropout=0.0, mlp_ratio=4)
    pp.load_state_dict({k[len("pick_place."):]: torch.from_numpy(v) for k, v in fx["state"].items()}, strict=True)
    pp = pp.to(gpu)
    pp.fusion.compute_dtype = torch.float32 if dtype == "fp32" else torch.bfloat16
    ins = [torch.from_numpy(fx["in"][k]).to(gpu).requires_grad_(True) for k in ("x_text", "x_ctx", "x_img")]
    leaves = [t.clone() for t in ins]                       # the type embedding is added in place (fusion.py:52)
    out = pp(*leaves, attention_masks=torch.from_numpy(fx["in"]["mask"]).to(gpu), modalities=[0, 1, 1])
    assert out["attn_weights"] is None
    worst = max(rel(out[k], ref) for k, ref in fx["out"].items())
    observed(f"masked_component_{dtype}", heat_rel=worst)
    assert worst < tol
    # the in-place side effect of the reference, and a working backward to inputs and parameters
    tw = pp.fusion.token_type_embeddings.weight
    assert torch.allclose(leaves[0], ins[0].detach() + tw[0].detach())
    sum(out[k].sum() for k in fx["out"]).backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0 for t in ins)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pp.parameters())
    # fusion alone
    feats, attn = pp.fusion(*[t.detach().clone() for t in ins], attention_masks=torch.from_numpy(fx["in"]["mask"]).to(gpu),
                            modalities=[0, 1, 1])
    assert attn is None and feats.shape == (2, 10, 96)


@pytest.mark.parametrize("detach", [False, True])
def test_standalone_component_with_the_mask_head_matches_the_oracle(gpu, observed, detach):
    """Components.get_by_name("pick_place_convdecoder", compute_mask=True) called on its own (models/pickplace.py:59-124):
    outputs, the gradients of both inputs and of every parameter -- the mask head's token gradient reaches the last input
    AND the type embedding the fusion model added to it in place (models/fusion.py:52) -- against the pinned restatement
    under autograd.  fp32 mode."""
    from oracle import bifold_oracle as O
    from bifold.models import Components
    torch.manual_seed(11 + detach)
    pp = Components.get_by_name("pick_place_convdecoder", dim=96, is_bimanual=True, fusion_model="concat_transformer",
                                num_patches=9, patch_size=16, heads=3, depth=2, dropout=0.0, mlp_ratio=4,
                                compute_mask=True, detach_mask=detach)
    state = {"pick_place." + k: v.detach().clone().requires_grad_(True) for k, v in pp.state_dict().items()}
    pp = pp.to(gpu)
    pp.fusion.compute_dtype = torch.float32
    g = torch.Generator().manual_seed(3)
    cpu_ins = [torch.randn(2, 5, 96, generator=g).requires_grad_(True), torch.randn(2, 10, 96, generator=g).requires_grad_(True)]
    ins = [t.detach().to(gpu).requires_grad_(True) for t in cpu_ins]
    out = pp(*[t.clone() for t in ins])
    ref = O.pick_place_convdecoder(state, cpu_ins, heads=3, is_bimanual=True, detach_mask=detach)
    keys = [k for k in ref if ref[k] is not None]
    assert set(keys) == {k for k in out if out[k] is not None} and "mask_heatmap" in keys
    worst = max(rel(out[k], ref[k].detach()) for k in keys)
    assert worst < 1e-4
    # (positive weights: the last conv's bias gradient is ONE number, the sum of a head's logit gradients -- signed
    # weights make it a cancelling sum whose fp32 summation order shows at 1e-3)
    wts = {k: torch.rand(ref[k].shape, generator=g) + 0.5 for k in keys}
    sum((ref[k] * wts[k]).sum() for k in keys).backward()
    sum((out[k] * wts[k].to(gpu)).sum() for k in keys).backward()
    seen = dict(out=worst, d_in=max(rel(a.grad, b.grad) for a, b in zip(ins, cpu_ins)))
    named = dict(pp.named_parameters())
    per = {k: rel(named[k[len("pick_place."):]].grad, v.grad) for k, v in state.items()}
    seen["d_param"] = max(per.values())
    seen["worst_param"] = max(per, key=per.get)
    observed(f"standalone_mask_head_detach{int(detach)}", **seen)
    assert seen["d_in"] < 2e-4 and seen["d_param"] < 2e-4
    assert rel(named["fusion.token_type_embeddings.weight"].grad, state["pick_place.fusion.token_type_embeddings.weight"].grad) < 2e-4


def _fusion_stack_both_ways(gpu, knob, cdt):
    """Outputs and every gradient of a 3-block, 256-wide fusion stack with pickplace.<knob> = 0 and = 1."""
    from bifold.models import Components
    from bifold.models import pickplace as P
    torch.manual_seed(5)
    fus = Components.get_by_name("concat_transformer", dim=256, heads=4, depth=3, dropout=0.0, mlp_ratio=4).to(gpu)
    fus.compute_dtype = cdt
    x_text = torch.randn(3, 5, 256, device=gpu)
    x_img = torch.randn(3, 17, 256, device=gpu)
    w = torch.randn(3, 17, 256, device=gpu)
    res = {}
    saved = getattr(P, knob)
    try:
        for mode in (0, 1):
            setattr(P, knob, mode)
            ins = [x_text.clone().requires_grad_(True), x_img.clone().requires_grad_(True)]
            fus.zero_grad(set_to_none=True)
            feats, _ = fus(ins[0].clone(), ins[1].clone())      # (the type embedding is added to the arguments in place)
            (feats * w).sum().backward()
            grads = {n: p.grad.detach().clone() for n, p in fus.named_parameters()}
            grads.update({f"input {i}": t.grad.detach().clone() for i, t in enumerate(ins)})
            res[mode] = (feats.detach().clone(), grads)
    finally:
        setattr(P, knob, saved)
    return rel(res[1][0], res[0][0]), max(rel(res[1][1][n], g) for n, g in res[0][1].items())


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_fusion_residual_add_inside_the_layernorm_pass_matches_the_epilogue_form(gpu, observed, dtype):
    """16-bit modes add the residual branches of a fusion block in the NEXT LayerNorm's pass (pickplace._ADD_IN_LN, the
    towers' form) instead of in the GEMM's fp32 epilogue.  Same function: outputs, input gradients and every parameter
    gradient of a 3-block stack must agree with the epilogue form to the rounding of ONE 16-bit value per branch
    (reference: transformer.py:19-28 `x = attn(x) + x; x = ff(x) + x`)."""
    tol = 2e-3 if dtype == "fp16" else 1.6e-2
    worst_out, worst_g = _fusion_stack_both_ways(gpu, "_ADD_IN_LN", torch.float16 if dtype == "fp16" else torch.bfloat16)
    observed(f"fusion_add_in_ln_{dtype}", out_rel=worst_out, grad_rel=worst_g)
    assert worst_out < tol and worst_g < 4 * tol


def test_fusion_backward_with_a_16_bit_gradient_stream_matches_the_fp32_stream(gpu, observed):
    """fp16 mode carries the residual stream's gradient through the fusion backward as one fp16 tensor
    (pickplace._GRAD_STREAM16: bf_layernorm_bwd without an fp32 dx).  The forward is untouched (identical outputs); every
    gradient agrees with the fp32 stream's to a few fp16 roundings of the running sum -- the gate the 16-bit gradients are
    held to everywhere else in this suite is 1e-2."""
    worst_out, worst_g = _fusion_stack_both_ways(gpu, "_GRAD_STREAM16", torch.float16)
    observed("fusion_grad_stream16", out_rel=worst_out, grad_rel=worst_g)
    assert worst_out == 0.0 and worst_g < 4e-3


def test_fp8_towers_accuracy_gate_on_config_a(gpu, observed):
    """BASELINE configs[4] (fp8 MFMA weights / activations): `dtype: fp8` runs the frozen towers' QKV / fc1 GEMMs in
    OCP e4m3 (per-token and per-channel scales).  There is no reference behaviour for fp8 (the reference is fp32), so
    the gate is stated here: on full-size config A the heatmaps stay within 5e-3 relative of the fp32 CPU oracle, the
    vision tower's tokens within 6e-2 of their absmax and the text tower's within 0.2 (observed on MI355X: heatmaps
    1.2e-3, vision 2.2e-2, text 8.0e-2, against 2.5e-4 / 4.5e-3 / 7.9e-3 for bf16 -- DESIGN.md §3): 5-10x bf16's
    error, which is why bf16 stays the headline dtype."""
    from oracle import bifold_oracle as O
    from bifold.models import Models
    # B = 96: the vision tower's GEMMs have M = 96 x 197 = 18 912 >= 16 384 rows, i.e. they dispatch the PERSISTENT e4m3
    # kernel (gemm_fp8_p8_kernel, csrc/gemm.hip) end to end; 8 of the 96 samples are checked against the CPU oracle
    B, N_CHECK = 96, 8
    gen = torch.Generator().manual_seed(3407)
    state = {**O.make_clip_state(gen), **O.make_tail_state(gen)}
    sample = O.synthetic_sample(B, torch.Generator().manual_seed(3419))
    dev_sample = {k: v.to(gpu) for k, v in sample.items()}
    sub = {k: v[:N_CHECK] for k, v in sample.items()}
    with torch.no_grad():
        ref = O.rgb_clip_forward(state, sub, heads=16)
        ref_vis = O.clip_vision_tokens(state, sub["rgb"])
        ref_txt = O.clip_text_tokens(state, sub["instruction"])
    seen = {}
    for dtype in ("fp8", "bf16"):
        with pytest.warns(UserWarning):
            model = Models.get_by_name(config_a(dtype), device=gpu)
        model.load_state_dict(state, strict=False)
        model = model.to(gpu).eval()
        assert model.clip_encoder.fp8 == (dtype == "fp8")
        with torch.no_grad():
            out = model(dev_sample)
            vis = model.clip_encoder.encode_image_with_embeddings(dev_sample["rgb"], torch.bfloat16).float().view(B, 197, 768)
            txt = model.clip_encoder.encode_text_with_embeddings(dev_sample["instruction"], torch.bfloat16).view(B, 77, 512)
        seen[f"{dtype}_heat"] = max(rel(out[k][:N_CHECK], ref[k]) for k in ("pick_heatmap", "place_heatmap"))
        seen[f"{dtype}_vision"] = rel(vis[:N_CHECK], ref_vis)
        seen[f"{dtype}_text"] = rel(txt[:N_CHECK], ref_txt)
        del model
        torch.cuda.empty_cache()
    observed("fp8_towers_config_a", **seen)
    assert seen["fp8_heat"] < 5e-3 and seen["fp8_vision"] < 6e-2 and seen["fp8_text"] < 0.2
    assert seen["bf16_heat"] < 1e-3


def test_vitl14_336_full_size_matches_oracle(gpu, observed):
    """BASELINE configs[3] at full size (ViT-L/14 @336: 24 vision layers of width 1024, 577 tokens; text width 768;
    fusion width 768 over 655 tokens; 384-pixel heatmaps), B = 2, the model's own random init handed to the oracle:
    heatmaps of both modes against the fp32 CPU restatement.  (The reference's own config would fail on the 384 vs 336
    target size; targets are not involved here.)"""
    from oracle import bifold_oracle as O
    from bifold.models import Models
    B = 2
    g = torch.Generator().manual_seed(7)
    sample = {"rgb": torch.randn(B, 3, 336, 336, generator=g),
              "instruction": O.synthetic_sample(B, torch.Generator().manual_seed(8))["instruction"]}
    seen = {}
    state = None
    for dtype in ("fp32", "bf16"):
        cfg = dict(config_a(dtype), image_size=336, text_encoder="ViT-L/14@336px", patch_size=14)
        torch.manual_seed(11)
        with pytest.warns(UserWarning):
            model = Models.get_by_name(cfg, device=gpu)
        if state is None:
            state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            with torch.no_grad():
                ref = O.rgb_clip_forward(state, sample, heads=16)
        else:
            model.load_state_dict(state)
        model = model.to(gpu).eval()
        with torch.no_grad():
            out = model({k: v.to(gpu) for k, v in sample.items()})
        assert out["pick_heatmap"].shape == (B, 384, 384)
        seen[dtype] = max(rel(out[k], ref[k]) for k in ("pick_heatmap", "place_heatmap"))
        del model
        torch.cuda.empty_cache()
    observed("vitl14_336_full", **seen)
    assert seen["fp32"] < 1e-3 and seen["bf16"] < 1e-3        # observed 2.1e-7 / 2.2e-4


def test_siglip_sequential_full_width_matches_oracle(gpu, observed):
    """BASELINE configs[2]'s model at full width (SigLIP-base/16 towers: 12 layers of 768, tanh-GELU; fusion 8 x 768
    with 16 heads of d = 48; bimanual, 4 decoders) at 224 px with T = 2 context frames, one of them masked: a
    656-token fusion sequence with the query-row mask, i.e. the tiled forward attention at d = 48.  B = 2, frozen
    towers, the model's own random init handed to the oracle."""
    from oracle import bifold_oracle as O
    from bifold.models import Models
    B, T, S = 2, 2, 224
    g = torch.Generator().manual_seed(9)
    sample = {"rgb": torch.randn(B, 3, S, S, generator=g), "rgb_context": torch.randn(B, T, 3, S, S, generator=g),
              "instruction": torch.randint(0, 32000, (B, 64), generator=g),
              "context_attention_mask": torch.tensor([[1, 0], [1, 1]])}
    seen = {}
    state = None
    for dtype in ("fp32", "bf16"):
        cfg = dict(name="siglip_sequential", image_size=S, is_bimanual=True, requires_graph=False, patch_size=16,
                   automodel_name="google/siglip-base-patch16-224", dim=768, emb_dropout=0.0, lora=False, r=8, lora_alpha=32,
                   lora_dropout=0.01, target_modules=["q_proj", "v_proj"], threshold=0.5, text_encoder=None,
                   pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", depth=8, heads=16,
                   mlp_ratio=4, dropout=0.0, dtype=dtype, context_length=T)
        torch.manual_seed(13)
        with pytest.warns(UserWarning):
            model = Models.get_by_name(cfg, device=gpu)
        if state is None:
            state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            with torch.no_grad():
                ref = O.siglip_sequential_forward(state, sample, 12, 16, True)
        else:
            model.load_state_dict(state)
        model = model.to(gpu).eval()
        with torch.no_grad():
            out = model({k: v.to(gpu) for k, v in sample.items()})
        seen[dtype] = max(rel(out[k], ref[k]) for k in ref if k.endswith("_heatmap"))
        del model
        torch.cuda.empty_cache()
    observed("siglip_sequential_full_width", **seen)
    assert seen["fp32"] < 1e-3 and seen["bf16"] < 1.5e-3      # observed 2.2e-7 / 4.3e-4


def _training_step_vs_oracle(gpu, cfg_of, sample, ref_forward, bim, dtypes, seed):
    """fwd + loss + bwd of the build (one model per dtype, same init) against the fp32 CPU oracle under torch
    autograd on the identical weights and batch: heatmaps, loss, every trainable gradient."""
    from oracle import bifold_oracle as O
    from bifold.losses import Losses
    from bifold.models import Models
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=bim, mask_pick_heatmap=False))
    dev_sample = {k: v.to(gpu) for k, v in sample.items()}
    state, ref, seen = None, None, {}
    for dtype in dtypes:
        torch.manual_seed(seed)
        with pytest.warns(UserWarning):
            model = Models.get_by_name(cfg_of(dtype), device=gpu)
        if state is None:
            state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
            rstate = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in state.items()}
            rout = ref_forward(rstate, sample)
            rloss, _ = O.bce_gaussmap(rout, sample, bim)
            rgrads = dict(zip(trainable, torch.autograd.grad(rloss, [rstate[k] for k in trainable])))
            ref = dict(out={k: v.detach() for k, v in rout.items() if k.endswith("_heatmap")}, loss=rloss.item())
            del rout, rloss, rstate
        else:
            model.load_state_dict(state)
        model = model.to(gpu).train()
        out = model(dev_sample)
        loss, _ = loss_fn(out, dev_sample)
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        named = dict(model.named_parameters())
        gerr = {k: rel(named[k].grad, rgrads[k]) for k in trainable}
        worst = max(gerr, key=gerr.get)
        lg = {k: (torch.logit(out[k].detach().double().cpu().clamp(1e-300, 1 - 1e-16)),
                  torch.logit(ref["out"][k].double().clamp(1e-300, 1 - 1e-16))) for k in ref["out"]}
        seen[dtype] = dict(heat=max(rel(out[k], ref["out"][k]) for k in ref["out"]),
                           logit=max(rel(a, b) for a, b in lg.values()),
                           loss=abs(loss.item() - ref["loss"]) / abs(ref["loss"]),
                           grad=gerr[worst], worst_grad=worst, median_grad=float(np.median(list(gerr.values()))))
        del model, out, loss, named
        torch.cuda.empty_cache()
    return seen


# Gates of the full-width training-step tests.  north_star's bar -- 1e-3 relative on LOGITS and HEATMAPS against the
# CPU path -- is asserted as such for the fp32 (exact) mode and for fp16, the shipped default (observed at full size:
# logits 2.4e-4 / 7.4e-4, heatmaps <= 5e-5).  bf16 meets it on heatmaps and loss only: its logits carry the 2^-9
# operand rounding (observed 2.4e-3 / 7.0e-3) and are held to the bound 1e-2 here, i.e. KNOWN to miss the gate, which
# is why bf16 is not the default.  Gradients are not part of north_star's gate; the 16-bit
# modes' carry their operand rounding (bf16 2^-9, fp16 2^-11) through 8 layers of backward and are bounded at ~3x
# what was observed (1.0-1.3e-3 fp16, 5-7e-3 bf16) so that a regression shows.
STEP_GATES = {"fp32": dict(heat=1e-3, logit=1e-3, loss=1e-3, grad=1e-3),
              "bf16": dict(heat=1e-3, logit=1e-2, loss=1e-3, grad=3e-2),
              "fp16": dict(heat=1e-3, logit=1e-3, loss=1e-3, grad=1e-2)}


def test_siglip_sequential_384_t4_training_step_matches_oracle(gpu, observed):
    """BASELINE configs[2]'s model at its FULL size (reference conf/dataset/bimanual_sequential.yaml at 384 px, T = 4 as
    BASELINE names it): SigLIP-base/16 @384 towers (576 patches), T = 4 context frames two of them masked, a
    2 950-token fusion sequence with the query-row mask (models/siglip.py:85-137, models/pickplace.py:73-106) -- the
    streamed d = 48 attention backward, the tanh-GELU towers, the 4-decoder collapse backward at width 768 -- one
    training step (forward, BCE, backward) at B = 1 against the oracle under torch autograd."""
    from oracle import bifold_oracle as O
    B, T, S = 1, 4, 384
    g = torch.Generator().manual_seed(21)
    sample = {"rgb": torch.randn(B, 3, S, S, generator=g), "rgb_context": torch.randn(B, T, 3, S, S, generator=g),
              "instruction": torch.randint(0, 32000, (B, 64), generator=g),
              "context_attention_mask": torch.tensor([[1, 1, 0, 0]])}
    for hn in O.decoder_heads(True):
        sample[f"{hn}_heatmap"] = O.gaussian_heatmaps(torch.randint(0, S, (B, 2), generator=g), S)

    def cfg_of(dtype):
        return dict(name="siglip_sequential", image_size=S, is_bimanual=True, requires_graph=False, patch_size=16,
                    automodel_name="google/siglip-base-patch16-384", dim=768, emb_dropout=0.0, lora=False, r=8,
                    lora_alpha=32, lora_dropout=0.01, target_modules=["q_proj", "v_proj"], threshold=0.5,
                    text_encoder=None, pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer",
                    depth=8, heads=16, mlp_ratio=4, dropout=0.0, dtype=dtype, context_length=T)
    seen = _training_step_vs_oracle(gpu, cfg_of, sample, lambda st, sm: O.siglip_sequential_forward(st, sm, 12, 16, True),
                                    True, list(STEP_GATES), seed=23)
    for dtype, v in seen.items():
        observed(f"siglip_sequential_384_t4_step_{dtype}", **v)
    for dtype, gate in STEP_GATES.items():
        assert all(seen[dtype][k] < gate[k] for k in gate), (dtype, seen[dtype])


def test_vitl14_336_training_step_matches_oracle(gpu, observed):
    """BASELINE configs[3]'s model at full size (ViT-L/14 @336: 24 x 1024 vision tower, 577 tokens; width-768 text tower
    and fusion over 655 tokens; the decoder upsamples the 24 x 24 patch grid x16 -> 384-pixel heatmaps and targets) --
    one training step at B = 2 against the oracle under torch autograd."""
    from oracle import bifold_oracle as O
    B = 2
    sample = O.synthetic_sample(B, torch.Generator().manual_seed(32), size=336, heat_size=384)

    def cfg_of(dtype):
        return dict(config_a(dtype), image_size=336, text_encoder="ViT-L/14@336px", patch_size=14)
    seen = _training_step_vs_oracle(gpu, cfg_of, sample, lambda st, sm: O.rgb_clip_forward(st, sm, heads=16), False,
                                    list(STEP_GATES), seed=33)
    for dtype, v in seen.items():
        observed(f"vitl14_336_step_{dtype}", **v)
    for dtype, gate in STEP_GATES.items():
        assert all(seen[dtype][k] < gate[k] for k in gate), (dtype, seen[dtype])

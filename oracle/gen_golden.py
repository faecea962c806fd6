#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (import from
/root/reference) and cross-check the oracle restatement against it.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever see the resulting fixtures.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--skip-full]

What is stubbed, and why it is not on the arithmetic: `timm`
(import-time symbol for the out-of-scope TransformerDecoder), `torchvision.
transforms` and `ftfy` (CLIP image preprocessing factory / tokenizer text
cleaning), `open3d` and `torchvision.transforms.v2` (module-level imports of the data
package; Processor._make_gaussmap touches neither).  Pretrained weights cannot be downloaded, so `load` is replaced by a
random-init CLIP of the requested hyper-parameters: parity is a property of the
arithmetic, not of the weights, and the fixtures carry the weights explicitly.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import transformers  # noqa: F401,E402  (its lazy loader probes timm; import before stubbing)

for _name in ("timm", "timm.models", "timm.models.vision_transformer", "ftfy", "torchvision",
              "torchvision.transforms"):
    if _name not in sys.modules:
        try:
            __import__(_name)
        except Exception:
            sys.modules[_name] = types.ModuleType(_name)
if not hasattr(sys.modules["timm.models.vision_transformer"], "Block"):
    sys.modules["timm.models.vision_transformer"].Block = object
_tv = sys.modules["torchvision.transforms"]
for _n in ("CenterCrop", "Compose", "Normalize", "Resize", "ToTensor", "InterpolationMode"):
    if not hasattr(_tv, _n):
        setattr(_tv, _n, type(_n, (), {"BICUBIC": 3}))

# `peft` (LoRA) is absent here and imported at the top of models/siglip.py; only the
# lora=False path is exercised, which never touches these names.
if "peft" not in sys.modules:
    try:
        import peft  # noqa: F401
    except Exception:
        for _name in ("peft", "peft.tuners", "peft.tuners.lora"):
            sys.modules[_name] = types.ModuleType(_name)
        sys.modules["peft.tuners.lora"].LoraConfig = type("LoraConfig", (), {})
        sys.modules["peft.tuners.lora"].LoraModel = type("LoraModel", (), {})

# bifold.data.processor (for Processor._make_gaussmap only) imports open3d and torchvision's v2
# transforms at module level; neither is installed and neither is touched by _make_gaussmap.
for _name in ("open3d", "torchvision.transforms.v2"):
    if _name not in sys.modules:
        try:
            __import__(_name)
        except Exception:
            sys.modules[_name] = types.ModuleType(_name)
if not hasattr(_tv, "v2"):
    _tv.v2 = sys.modules["torchvision.transforms.v2"]

import bifold.models.rgb_clip as ref_rgb_clip  # noqa: E402
import bifold.models.siglip as ref_siglip  # noqa: E402
from bifold.losses import Losses as RefLosses  # noqa: E402
from bifold.data.processor import Processor as RefProcessor  # noqa: E402
from bifold.models import BaseModel as RefBaseModel  # noqa: E402
from bifold.models import Components as RefComponents  # noqa: E402
from bifold.models.utils import sample_from_heatmap as ref_sample_from_heatmap  # noqa: E402
from bifold.models.clip.model import CLIP as RefCLIP  # noqa: E402
from bifold.models.modules import ConvDecoder as RefConvDecoder  # noqa: E402
from bifold.optim import Optimizers as RefOptimizers  # noqa: E402
from bifold.optim import Schedulers as RefSchedulers  # noqa: E402

from oracle import bifold_oracle as O  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


class Cfg(dict):
    """Attribute-style dict standing in for an OmegaConf node (factories do
    `cfg.name`, `del cfg.name`, `**cfg`)."""
    __getattr__ = dict.__getitem__
    __delattr__ = dict.__delitem__


def np_state(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def check(name, a, b, rtol=2e-5, atol=2e-6):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    ok = err <= atol + rtol * ref
    print(f"  [{'ok' if ok else 'FAIL'}] {name}: max|d|={err:.3e} ref max={ref:.3e}")
    if not ok:
        raise SystemExit(f"oracle != reference for {name}")


# --------------------------------------------------------------------------
def build_ref_rgbonly(clip_kwargs, model_kwargs, seed):
    torch.manual_seed(seed)

    def fake_load(name, device="cpu", **_):
        return RefCLIP(**clip_kwargs).float().eval(), None

    ref_rgb_clip.load = fake_load
    model = ref_rgb_clip.RGBOnly(device=torch.device("cpu"), **model_kwargs)
    # the reference inits LayerNorm to (1, 0) and CLIP biases to default; perturb so that
    # scale/shift parameters are exercised by the fixtures
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if ("ln_" in k or ".norm." in k) and k.endswith("weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif ("ln_" in k or ".norm." in k) and k.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return model


def run_reference_training(model, sample, is_bimanual, steps, lr=1e-4):
    """Trainer.prepare_train + train_epoch semantics (bifold/__main__.py:112-163)
    driven through the reference's own factories."""
    loss_fn = RefLosses.get_by_name(cfg=Cfg(name="bce_gaussmap", is_bimanual=is_bimanual,
                                            mask_pick_heatmap=False))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = RefOptimizers.get_by_name(cfg=Cfg(name="adam", lr=lr, betas=[0.9, 0.999], eps=1e-8,
                                            weight_decay=0), params=params)
    model.train()
    rec = {}
    for it in range(steps):
        out = model({k: v.clone() for k, v in sample.items()})
        loss, parts = loss_fn(out, sample)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            rec["out"] = {k: v.detach().clone() for k, v in out.items() if v is not None}
            rec["loss"] = loss.detach().clone()
            rec["parts"] = {k: v.detach().clone() for k, v in parts.items()}
            rec["grads"] = {k: p.grad.detach().clone() for k, p in model.named_parameters()
                            if p.requires_grad}
        rec.setdefault("losses", []).append(float(loss))
        opt.step()
    rec["params_after"] = {k: p.detach().clone() for k, p in model.named_parameters()
                           if p.requires_grad}
    return rec


def oracle_training(state0, sample, heads, is_bimanual, steps, lr=1e-4):
    state = {k: v.clone() for k, v in state0.items()}
    train_keys = [k for k in state if O.is_trainable_key(k)]
    m = {k: torch.zeros_like(state[k]) for k in train_keys}
    v = {k: torch.zeros_like(state[k]) for k in train_keys}
    rec = {"losses": []}
    for it in range(steps):
        for k in train_keys:
            state[k] = state[k].detach().requires_grad_(True)
        out = O.rgb_clip_forward(state, sample, heads, is_bimanual)
        loss, parts = O.bce_gaussmap(out, sample, is_bimanual)
        grads = torch.autograd.grad(loss, [state[k] for k in train_keys])
        if it == 0:
            rec["out"] = {k: t.detach() for k, t in out.items() if t is not None}
            rec["loss"] = loss.detach()
            rec["grads"] = dict(zip(train_keys, grads))
        rec["losses"].append(float(loss))
        with torch.no_grad():
            for k, g in zip(train_keys, grads):
                state[k], m[k], v[k] = O.adam_update(state[k].detach(), g, m[k], v[k], it + 1, lr)
    rec["params_after"] = {k: state[k].detach() for k in train_keys}
    return rec


TINY_CLIP = dict(embed_dim=32, image_resolution=64, vision_layers=2, vision_width=128,
                 vision_patch_size=16, context_length=77, vocab_size=512, transformer_width=64,
                 transformer_heads=1, transformer_layers=2)
TINY_MODEL = dict(image_size=64, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
                  patch_size=16, depth=2, heads=2, mlp_ratio=4, dropout=0.0, text_dropout=0.0,
                  rgb_dropout=0.0, pick_place_model="pick_place_convdecoder",
                  fusion_model="concat_transformer")


def tiny_end_to_end(is_bimanual: bool, fname: str, seed: int, patch: int = 16):
    """patch = 14 is the ViT-L/14 geometry in miniature (BASELINE configs[3]): 56-pixel images, a 4x4
    patch grid, K = 588 in the patch embedding, and 64-pixel heatmaps (the decoder upsamples x16)."""
    print(f"== tiny end-to-end, bimanual={is_bimanual}, patch={patch}")
    size = 4 * patch
    mk = dict(TINY_MODEL, is_bimanual=is_bimanual, image_size=size, patch_size=patch)
    model = build_ref_rgbonly(dict(TINY_CLIP, image_resolution=size, vision_patch_size=patch), mk, seed)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(seed + 7)
    sample = O.synthetic_sample(3, gen, size=size, vocab=512, is_bimanual=is_bimanual, heat_size=64)

    # intermediates from the reference
    with torch.no_grad():
        x_text = model.encode_text(sample["instruction"]).clone()
        x_rgb = model.encode_rgb(sample["rgb"]).clone()
        vis = model.clip_encoder.encode_image_with_embeddings(sample["rgb"]).clone()
        txt = model.clip_encoder.encode_text_with_embeddings(sample["instruction"]).clone()
        fused, _ = model.pick_place.fusion(x_text.clone(), x_rgb.clone())
    rec = run_reference_training(model, sample, is_bimanual, steps=3)

    # oracle vs reference
    with torch.no_grad():
        check("clip vision tokens", O.clip_vision_tokens(state0, sample["rgb"]), vis)
        check("clip text tokens", O.clip_text_tokens(state0, sample["instruction"]), txt)
    orec = oracle_training(state0, sample, heads=mk["heads"], is_bimanual=is_bimanual, steps=3)
    for k, t in rec["out"].items():
        check(f"out[{k}]", orec["out"][k], t)
    check("loss", orec["loss"], rec["loss"], rtol=1e-6)
    for k, g in rec["grads"].items():
        check(f"grad[{k}]", orec["grads"][k], g, rtol=1e-4, atol=1e-7)
    for k, p in rec["params_after"].items():
        check(f"param3[{k}]", orec["params_after"][k], p, rtol=1e-5, atol=1e-6)

    blob = {}
    for k, t in np_state(state0).items():
        blob["state/" + k] = t
    for k, t in sample.items():
        blob["sample/" + k] = t.numpy()
    blob["mid/clip_vision"] = vis.numpy()
    blob["mid/clip_text"] = txt.numpy()
    blob["mid/x_text"] = x_text.numpy()
    blob["mid/x_rgb"] = x_rgb.numpy()
    blob["mid/fused"] = fused.numpy()
    for k, t in rec["out"].items():
        blob["out/" + k] = t.numpy()
    blob["loss"] = rec["loss"].numpy()
    blob["losses3"] = np.array(rec["losses"], dtype=np.float64)
    for k, t in rec["parts"].items():
        blob["loss_part/" + k] = t.numpy()
    for k, t in rec["grads"].items():
        blob["grad/" + k] = t.numpy()
    for k, t in rec["params_after"].items():
        blob["param3/" + k] = t.numpy()
    blob["meta/heads"] = np.array(mk["heads"])
    blob["meta/is_bimanual"] = np.array(int(is_bimanual))
    np.savez_compressed(os.path.join(GOLD, fname), **blob)
    print(f"  wrote {fname}")


# --------------------------------------------------------------------------
def mask_head_fixture(is_bimanual: bool, detach_mask: bool, loss_cfg: dict, fname: str, seed: int):
    """The reference's RGBOnly with `compute_mask=True` (models/pickplace.py:25-30,66-69,76-119) trained for three
    steps under one of the reference's own composed losses (conf/loss/composed.yaml, composed_dice_focal.yaml:
    losses/__init__.py:30-51,112-155) -- and every single loss evaluated on the first step's output."""
    print(f"== mask head, bimanual={is_bimanual}, detach_mask={detach_mask}, loss={loss_cfg['loss_names']}")
    mk = dict(TINY_MODEL, is_bimanual=is_bimanual, compute_mask=True, detach_mask=detach_mask)
    model = build_ref_rgbonly(dict(TINY_CLIP), mk, seed)
    assert model.pick_place.mask_head is not None
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(seed + 7)
    sample = O.synthetic_sample(3, gen, size=64, vocab=512, is_bimanual=is_bimanual, heat_size=64)
    # a binary cloth mask [B, 1, S, S] (what the datasets hand over): a thresholded smooth field, one sample all zero
    field = torch.nn.functional.interpolate(torch.randn(3, 1, 8, 8, generator=gen), size=64, mode="bilinear")
    sample["mask"] = (field > 0.1).float()
    sample["mask"][2] = 0.0
    ref_loss = RefLosses.get_by_name(cfg=Cfg(name="composed", is_bimanual=is_bimanual, **loss_cfg))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = RefOptimizers.get_by_name(cfg=Cfg(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                                    params=params)
    model.train()
    rec = {"losses": []}
    for it in range(3):
        out = model({k: v.clone() for k, v in sample.items()})
        loss, parts = ref_loss(out, sample)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            rec["out"] = {k: v.detach().clone() for k, v in out.items() if v is not None}
            rec["loss"] = loss.detach().clone()
            rec["parts"] = {k: v.detach().clone() for k, v in parts.items()}
            rec["grads"] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
            with torch.no_grad():
                rec["single"] = {n: RefLosses.get_by_name(name=n, is_bimanual=is_bimanual,
                                                          mask_pick_heatmap=loss_cfg["mask_pick_heatmap"])(out, sample)[0]
                                 .detach().clone() for n in ("bce_gaussmap", "bce_mask", "dice", "focal")}
        rec["losses"].append(float(loss))
        opt.step()
    rec["params_after"] = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}

    # the restatement against the run
    state = {k: v.clone() for k, v in state0.items()}
    train_keys = [k for k in state if O.is_trainable_key(k)]
    m = {k: torch.zeros_like(state[k]) for k in train_keys}
    v = {k: torch.zeros_like(state[k]) for k in train_keys}
    for it in range(3):
        for k in train_keys:
            state[k] = state[k].detach().requires_grad_(True)
        out = O.rgb_clip_forward(state, sample, mk["heads"], is_bimanual, detach_mask=detach_mask)
        loss, parts = O.composed_loss(out, sample, is_bimanual=is_bimanual, **loss_cfg)
        grads = torch.autograd.grad(loss, [state[k] for k in train_keys], allow_unused=True)
        if it == 0:
            for k, t in rec["out"].items():
                check(f"out[{k}]", out[k].detach(), t)
            check("loss", loss.detach(), rec["loss"], rtol=1e-6)
            for k, t in rec["parts"].items():
                check(f"part[{k}]", parts[k].detach(), t, rtol=1e-6)
            single = {"bce_gaussmap": O.bce_gaussmap(out, sample, is_bimanual, loss_cfg["mask_pick_heatmap"])[0],
                      "bce_mask": O.bce_mask(out, sample)[0], "dice": O.dice_loss(out, sample)[0],
                      "focal": O.focal_loss(out, sample)[0]}
            for k, t in rec["single"].items():
                check(f"single[{k}]", single[k].detach(), t, rtol=1e-6)
            for k, g in zip(train_keys, grads):
                check(f"grad[{k}]", torch.zeros_like(state[k]) if g is None else g, rec["grads"][k], rtol=1e-4, atol=1e-7)
        assert abs(float(loss) - rec["losses"][it]) <= 1e-5 * abs(rec["losses"][it])
        with torch.no_grad():
            for k, g in zip(train_keys, grads):
                g = torch.zeros_like(state[k]) if g is None else g
                state[k], m[k], v[k] = O.adam_update(state[k].detach(), g, m[k], v[k], it + 1, 1e-4)
    # (atol: an element whose gradient is of the size of Adam's eps = 1e-8 moves by lr * g / (|g| + eps) -- fp32 noise
    # of 1e-9 in such a gradient is 5e-6 of movement at lr = 1e-4; the all-zero mask of sample 2 makes such elements)
    for k, p_ in rec["params_after"].items():
        check(f"param3[{k}]", state[k].detach(), p_, rtol=1e-5, atol=2e-5)

    blob = {}
    for k, t in np_state(state0).items():
        blob["state/" + k] = t
    for k, t in sample.items():
        blob["sample/" + k] = t.numpy()
    for k, t in rec["out"].items():
        blob["out/" + k] = t.numpy()
    blob["loss"] = rec["loss"].numpy()
    blob["losses3"] = np.array(rec["losses"], dtype=np.float64)
    for k, t in rec["parts"].items():
        blob["loss_part/" + k] = t.numpy()
    for k, t in rec["single"].items():
        blob["single/" + k] = t.numpy()
    for k, t in rec["grads"].items():
        blob["grad/" + k] = t.numpy()
    for k, t in rec["params_after"].items():
        blob["param3/" + k] = t.numpy()
    blob["meta/heads"] = np.array(mk["heads"])
    blob["meta/is_bimanual"] = np.array(int(is_bimanual))
    blob["meta/detach_mask"] = np.array(int(detach_mask))
    blob["meta/loss_names"] = np.array(loss_cfg["loss_names"])
    blob["meta/weights"] = np.array(loss_cfg["weights"], dtype=np.float64)
    blob["meta/mask_pick_heatmap"] = np.array(int(loss_cfg["mask_pick_heatmap"]))
    np.savez_compressed(os.path.join(GOLD, fname), **blob)
    print(f"  wrote {fname}")


MASK_LOSS_DICE_FOCAL = dict(loss_names=["bce_gaussmap", "focal", "dice"], weights=[40, 20, 1], mask_pick_heatmap=True)
MASK_LOSS_BCE = dict(loss_names=["bce_gaussmap", "bce_mask"], weights=[2, 1], mask_pick_heatmap=True)


def mask_head_fixtures():
    mask_head_fixture(False, False, MASK_LOSS_DICE_FOCAL, "tiny_mask_head.npz", seed=3417)
    mask_head_fixture(True, True, MASK_LOSS_BCE, "tiny_mask_head_bimanual_detached.npz", seed=3418)



def masked_fusion_fixture(seed: int):
    """PickPlaceConvDecoder driven the way SiglipSequential drives it
    (models/siglip.py:118-137): three inputs, modalities [0,1,1], a query-row
    attention mask with zeros, bimanual (4 decoders)."""
    print("== masked 3-input fusion + 4 decoders")
    torch.manual_seed(seed)
    dim, heads, depth, g = 96, 2, 2, 3                      # d_head = 48 like the Q config
    pp = RefComponents.get_by_name("pick_place_convdecoder", dim=dim, is_bimanual=True,
                                   fusion_model="concat_transformer", num_patches=g * g,
                                   patch_size=16, heads=heads, depth=depth, dropout=0.0,
                                   mlp_ratio=4)
    gen = torch.Generator().manual_seed(seed + 3)
    B, T = 2, 2
    n_img = g * g + 1
    x_text = torch.randn(B, 5, dim, generator=gen)
    x_ctx = torch.randn(B, T * n_img, dim, generator=gen)
    x_img = torch.randn(B, n_img, dim, generator=gen)
    ctx_mask = torch.tensor([[1, 0], [0, 0]], dtype=torch.int64)
    mask = torch.cat([torch.ones(B, 5, dtype=torch.int64),
                      ctx_mask.repeat_interleave(n_img, dim=1),
                      torch.ones(B, n_img, dtype=torch.int64)], dim=1)
    out = pp(x_text.clone(), x_ctx.clone(), x_img.clone(), attention_masks=mask,
             modalities=[0, 1, 1])
    state = {"pick_place." + k: v.detach() for k, v in pp.state_dict().items()}
    with torch.no_grad():
        oo = O.pick_place_convdecoder(state, [x_text, x_ctx, x_img], heads, True,
                                      modalities=[0, 1, 1], attention_masks=mask)
    blob = {"in/x_text": x_text.numpy(), "in/x_ctx": x_ctx.numpy(), "in/x_img": x_img.numpy(),
            "in/mask": mask.numpy(), "meta/heads": np.array(heads)}
    for k, t in out.items():
        if t is not None:
            check(f"masked out[{k}]", oo[k], t)
            blob["out/" + k] = t.detach().numpy()
    for k, t in np_state(state).items():
        blob["state/" + k] = t
    np.savez_compressed(os.path.join(GOLD, "fusion_masked_bimanual.npz"), **blob)
    print("  wrote fusion_masked_bimanual.npz")


def per_op_fixture(seed: int):
    print("== per-op vectors")
    gen = torch.Generator().manual_seed(seed)
    blob = {}
    # LayerNorm at the real widths
    for w in (512, 768):
        ln = torch.nn.LayerNorm(w)
        with torch.no_grad():
            ln.weight.copy_(1 + 0.2 * torch.randn(w, generator=gen))
            ln.bias.copy_(0.2 * torch.randn(w, generator=gen))
        x = 3.0 * torch.randn(5, w, generator=gen) + 0.5
        y = ln(x).detach()
        check(f"layernorm {w}", O.layer_norm(x, ln.weight.detach(), ln.bias.detach()), y)
        blob[f"ln{w}/x"], blob[f"ln{w}/w"], blob[f"ln{w}/b"], blob[f"ln{w}/y"] = (
            x.numpy(), ln.weight.detach().numpy(), ln.bias.detach().numpy(), y.numpy())
    # activations
    from bifold.models.clip.model import QuickGELU as RefQuickGELU
    x = torch.linspace(-8, 8, 257)
    blob["act/x"] = x.numpy()
    blob["act/quick_gelu"] = RefQuickGELU()(x).numpy()
    blob["act/erf_gelu"] = torch.nn.GELU()(x).numpy()
    check("quick_gelu", O.quick_gelu(x), blob["act/quick_gelu"])
    check("erf_gelu", O.erf_gelu(x), blob["act/erf_gelu"])
    # one CLIP text block (causal MHA, 77 tokens, width 128 -> 2 heads of 64)
    from bifold.models.clip.model import ResidualAttentionBlock as RefBlock
    torch.manual_seed(seed)
    mask = torch.empty(77, 77).fill_(float("-inf")).triu_(1)
    blk = RefBlock(128, 2, mask).eval()
    x = torch.randn(2, 77, 128, generator=gen)
    with torch.no_grad():
        y = blk(x.permute(1, 0, 2)).permute(1, 0, 2).contiguous()
    st = {"b." + k: v.detach() for k, v in blk.state_dict().items()}
    with torch.no_grad():
        check("clip causal block", O.clip_resblock(st, "b.", x, 2, causal=True), y)
    blob["clipblock/x"], blob["clipblock/y"] = x.numpy(), y.numpy()
    for k, v in st.items():
        blob["clipblock/state/" + k] = v.numpy()
    # ConvDecoder at D=512, g=14 (full size), B=1: as-written vs collapsed form
    torch.manual_seed(seed + 1)
    dec = RefConvDecoder(512).eval()
    feat = torch.randn(1, 512, 14, 14, generator=gen)
    with torch.no_grad():
        logits = dec(feat)
    st = {"d." + k: v.detach() for k, v in dec.state_dict().items()}
    with torch.no_grad():
        check("conv decoder as written", O.conv_decoder_logits(st, "d.", feat), logits)
    w_eff, b_eff = O.collapse_decoder(st, "d.")
    U = O.upsample_operator(14)
    low = torch.einsum("d,bdij->bij", w_eff, feat.double()) + b_eff
    coll = torch.einsum("yi,bij,xj->byx", U, low, U)
    check("conv decoder collapsed", coll, logits.squeeze(1), rtol=1e-5, atol=1e-6)
    blob["decoder/feat"], blob["decoder/logits"] = feat.numpy(), logits.numpy()
    blob["decoder/upsample_operator_14"] = U.numpy()
    for k, v in st.items():
        blob["decoder/state/" + k] = v.numpy()
    # BCE incl. the clamp edge (p exactly 0 / 1)
    p = torch.tensor([0.0, 1.0, 1e-30, 0.5, 0.999999, 0.25, 1.0, 0.0])
    t = torch.tensor([1.0, 0.0, 1.0, 0.3, 1.0, 0.0, 1.0, 0.0])
    l = torch.nn.BCELoss()(p, t)
    check("bce clamp", O.bce_mean(p, t), l)
    blob["bce/p"], blob["bce/t"], blob["bce/loss"] = p.numpy(), t.numpy(), l.numpy()
    # LinearWarmup LR trace (15 optimiser steps, 100 iters, base 1e-4)
    prm = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.Adam(prm, lr=1e-4)
    sch = RefSchedulers.get_by_name(Cfg(name="linear_warmup", warmup_portion=0.1, warmup_start_lr=0,
                                        use_cosine_decay=True), optimizer=opt, max_iters=100)
    trace = []
    for _ in range(30):
        trace.append(opt.param_groups[0]["lr"])
        prm[0].grad = torch.ones(1)
        opt.step()
        sch.step()
    mine = O.linear_warmup_lrs(1e-4, 100, 30)
    check("linear_warmup trace", torch.tensor(mine), torch.tensor(trace), rtol=1e-9, atol=1e-15)
    blob["warmup/lrs"] = np.array(trace, dtype=np.float64)
    # Adam: 4 steps on a fixed gradient sequence
    pa = torch.nn.Parameter(torch.randn(64, generator=gen))
    p0 = pa.detach().clone()
    opt = torch.optim.Adam([pa], lr=1e-3)
    gs = torch.randn(4, 64, generator=gen)
    for i in range(4):
        pa.grad = gs[i].clone()
        opt.step()
    q, m, v = p0.clone(), torch.zeros(64), torch.zeros(64)
    for i in range(4):
        q, m, v = O.adam_update(q, gs[i], m, v, i + 1, 1e-3)
    check("adam 4 steps", q, pa.detach(), rtol=1e-6, atol=1e-7)
    blob["adam/p0"], blob["adam/grads"], blob["adam/p4"] = p0.numpy(), gs.numpy(), pa.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "per_op.npz"), **blob)
    print("  wrote per_op.npz")


TINY_SIGLIP = dict(text_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                    num_attention_heads=2, vocab_size=100, max_position_embeddings=64),
                   vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                      num_attention_heads=2, image_size=32, patch_size=16))


def siglip_fixture(sequential: bool, fname: str, seed: int):
    """SigLip / SiglipSequential (models/siglip.py) with lora=False around a random-init
    transformers.SiglipModel: the towers are third-party code, run here as the reference runs it."""
    from transformers import SiglipConfig, SiglipModel
    print(f"== tiny siglip, sequential={sequential}")
    torch.manual_seed(seed)

    class FakeAuto:
        @staticmethod
        def from_pretrained(name):
            m = SiglipModel(SiglipConfig(**TINY_SIGLIP)).float().eval()
            g = torch.Generator().manual_seed(seed + 5)
            with torch.no_grad():                      # exercise LayerNorm scale/shift and biases
                for k, p in m.named_parameters():
                    if "layer_norm" in k or "layernorm" in k or k.endswith("bias"):
                        p.add_(0.1 * torch.randn(p.shape, generator=g))
            return m

    ref_siglip.AutoModel = FakeAuto
    is_bimanual = sequential                      # the reference pairs sequential with bimanual data
    kw = dict(device=torch.device("cpu"), image_size=32, is_bimanual=is_bimanual, requires_graph=False,
              patch_size=16, dim=64, lora=False, r=8, lora_alpha=32, lora_dropout=0.01, automodel_name="tiny",
              target_modules=["q_proj", "v_proj"], pick_place_model="pick_place_convdecoder",
              fusion_model="concat_transformer", depth=2, heads=2, mlp_ratio=4, dropout=0.0, emb_dropout=0.0,
              threshold=0.5, text_encoder=None)
    T = 2
    model = (ref_siglip.SiglipSequential(context_length=T, **kw) if sequential else ref_siglip.SigLip(**kw))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if ".norm." in k:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B = 3
    gen = torch.Generator().manual_seed(seed + 7)
    sample = {"rgb": torch.randn(B, 3, 32, 32, generator=gen),
              "instruction": torch.randint(0, 100, (B, 8), generator=gen)}
    if sequential:
        sample["rgb_context"] = torch.randn(B, T, 3, 32, 32, generator=gen)
        sample["context_attention_mask"] = torch.tensor([[1, 0], [1, 1], [0, 0]], dtype=torch.int64)
    for n in O.decoder_heads(is_bimanual):
        c = torch.randint(0, 32, (B, 2), generator=gen)
        sample[f"{n}_heatmap"] = O.gaussian_heatmaps(c, 32, sigma=3.0)
    with torch.no_grad():
        out = model.siglip_model(input_ids=sample["instruction"], pixel_values=sample["rgb"])
        vis = out.vision_model_output.last_hidden_state.clone()
        txt = out.text_model_output.last_hidden_state.clone()
        check("siglip vision tokens", O.siglip_vision_tokens(state0, sample["rgb"], 2), vis)
        check("siglip text tokens", O.siglip_text_tokens(state0, sample["instruction"], 2), txt)
    assert all(not p.requires_grad for k, p in model.named_parameters() if "siglip_model" in k)
    rec = run_reference_training(model, sample, is_bimanual, steps=3)
    fwd = O.siglip_sequential_forward if sequential else O.siglip_forward
    state = {k: v.clone() for k, v in state0.items()}
    keys = [k for k in state if "siglip_model" not in k]
    for k in keys:
        state[k].requires_grad_(True)
    oo = fwd(state, sample, 2, 2, is_bimanual)
    loss, _ = O.bce_gaussmap(oo, sample, is_bimanual)
    grads = dict(zip(keys, torch.autograd.grad(loss, [state[k] for k in keys])))
    for k, t in rec["out"].items():
        check(f"out[{k}]", oo[k].detach(), t)
    check("loss", loss.detach(), rec["loss"], rtol=1e-6)
    assert set(rec["grads"]) == set(keys), set(rec["grads"]) ^ set(keys)
    for k, gr in rec["grads"].items():
        check(f"grad[{k}]", grads[k], gr, rtol=1e-4, atol=1e-7)
    blob = {}
    for k, t in np_state(state0).items():
        blob["state/" + k] = t
    for k, t in sample.items():
        blob["sample/" + k] = t.numpy()
    blob["mid/siglip_vision"], blob["mid/siglip_text"] = vis.numpy(), txt.numpy()
    for k, t in rec["out"].items():
        blob["out/" + k] = t.numpy()
    blob["loss"] = rec["loss"].numpy()
    blob["losses3"] = np.array(rec["losses"], dtype=np.float64)
    for k, t in rec["grads"].items():
        blob["grad/" + k] = t.numpy()
    for k, t in rec["params_after"].items():
        blob["param3/" + k] = t.numpy()
    blob["meta/is_bimanual"] = np.array(int(is_bimanual))
    blob["meta/context_length"] = np.array(T if sequential else 0)
    np.savez_compressed(os.path.join(GOLD, fname), **blob)
    print(f"  wrote {fname}")


def siglip_lora_fixture(sequential: bool, fname: str, seed: int, r: int = 8, lora_alpha: float = 32.0):
    """A reference-run pin for `lora: true` (models/siglip.py:33-40) without `peft`.

    `peft` is absent and is neither imported nor stood in for.  What is assumed about it is ONLY the defining
    formula of a LoRA-wrapped Linear at dropout 0, W' = W + (alpha / r) B A (peft.tuners.lora: `merge`), and PEFT's key
    naming.  Everything else is the reference itself: its SigLip / SiglipSequential with `lora=False` is built around
    towers whose q_proj / v_proj weights ARE W', those weights get `requires_grad` back, and the reference's own
    forward, loss and autograd give heatmaps, loss, every tail gradient and G = dL/dW'.  By the chain rule on the
    formula the adapter gradients are dA = (alpha / r) B^T G and dB = (alpha / r) G A^T: the tests derive them from
    the recorded G.  The same run pins oracle/lora_oracle.py at dropout 0 (asserted here)."""
    from transformers import SiglipConfig, SiglipModel
    from oracle import lora_oracle as LO
    print(f"== tiny siglip with merged LoRA weights, sequential={sequential}")
    torch.manual_seed(seed)
    scale = lora_alpha / r
    adapters = {}

    class FakeAuto:
        @staticmethod
        def from_pretrained(name):
            m = SiglipModel(SiglipConfig(**TINY_SIGLIP)).float().eval()
            g = torch.Generator().manual_seed(seed + 5)
            with torch.no_grad():
                for k, p in m.named_parameters():
                    if "layer_norm" in k or "layernorm" in k or k.endswith("bias"):
                        p.add_(0.1 * torch.randn(p.shape, generator=g))
                for k, p in m.named_parameters():
                    if k.endswith(("self_attn.q_proj.weight", "self_attn.v_proj.weight")) and ".head." not in k:
                        a = torch.randn(r, p.shape[1], generator=g) * p.shape[1] ** -0.5
                        b = torch.randn(p.shape[0], r, generator=g) * 0.05
                        adapters[k] = (p.detach().clone(), a, b)
                        p.add_(scale * (b @ a))
            return m

    ref_siglip.AutoModel = FakeAuto
    is_bimanual = sequential
    kw = dict(device=torch.device("cpu"), image_size=32, is_bimanual=is_bimanual, requires_graph=False,
              patch_size=16, dim=64, lora=False, r=r, lora_alpha=lora_alpha, lora_dropout=0.0, automodel_name="tiny",
              target_modules=["q_proj", "v_proj"], pick_place_model="pick_place_convdecoder",
              fusion_model="concat_transformer", depth=2, heads=2, mlp_ratio=4, dropout=0.0, emb_dropout=0.0,
              threshold=0.5, text_encoder=None)
    T = 2
    model = (ref_siglip.SiglipSequential(context_length=T, **kw) if sequential else ref_siglip.SigLip(**kw))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if ".norm." in k:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    assert len(adapters) == 2 * 2 * 2                     # towers x layers x {q, v}
    named = dict(model.named_parameters())
    for k in adapters:
        named["siglip_model." + k].requires_grad_(True)   # the only tower tensors LoRA lets a gradient reach
    B = 3
    gen = torch.Generator().manual_seed(seed + 7)
    sample = {"rgb": torch.randn(B, 3, 32, 32, generator=gen),
              "instruction": torch.randint(0, 100, (B, 8), generator=gen)}
    if sequential:
        sample["rgb_context"] = torch.randn(B, T, 3, 32, 32, generator=gen)
        sample["context_attention_mask"] = torch.tensor([[1, 0], [1, 1], [0, 0]], dtype=torch.int64)
    for n in O.decoder_heads(is_bimanual):
        c = torch.randint(0, 32, (B, 2), generator=gen)
        sample[f"{n}_heatmap"] = O.gaussian_heatmaps(c, 32, sigma=3.0)
    loss_fn = RefLosses.get_by_name(cfg=Cfg(name="bce_gaussmap", is_bimanual=is_bimanual, mask_pick_heatmap=False))
    model.train()
    out = model({k: v.clone() for k, v in sample.items()})
    loss, _ = loss_fn(out, sample)
    loss.backward()
    out = {k: v.detach() for k, v in out.items() if v is not None}

    # the state a `lora: true` model carries, in PEFT's key layout
    state = {}
    for k, v in model.state_dict().items():
        if not k.startswith("siglip_model."):
            state[k] = v.detach().clone()
            continue
        hf = k[len("siglip_model."):]
        stem = hf.rsplit(".", 1)[0]
        if stem + ".weight" in adapters:
            state[LO.PEFT_PREFIX + stem + ".base_layer." + hf.rsplit(".", 1)[1]] = (
                adapters[stem + ".weight"][0] if hf.endswith(".weight") else v.detach().clone())
        else:
            state[LO.PEFT_PREFIX + hf] = v.detach().clone()
    for k, (_, a, b) in adapters.items():
        stem = k.rsplit(".", 1)[0]
        state[f"{LO.PEFT_PREFIX}{stem}.lora_A.siglip_adapter.weight"] = a
        state[f"{LO.PEFT_PREFIX}{stem}.lora_B.siglip_adapter.weight"] = b
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()
             if p.requires_grad and not k.startswith("siglip_model.")}
    grad_w = {k: named["siglip_model." + k].grad.detach().clone() for k in adapters}

    # pin oracle/lora_oracle.py (adapters applied unmerged, torch autograd) at dropout 0 against this run
    lo_state = {k: v.clone() for k, v in state.items()}
    keys = list(LO.trainable_keys(lo_state))
    for k in keys:
        lo_state[k].requires_grad_(True)
    fwd = LO.siglip_sequential_forward if sequential else LO.siglip_forward
    oo = fwd(lo_state, sample, 2, 2, is_bimanual, r=r, lora_alpha=lora_alpha)
    lo_loss, _ = O.bce_gaussmap(oo, sample, is_bimanual)
    lo_grads = dict(zip(keys, torch.autograd.grad(lo_loss, [lo_state[k] for k in keys])))
    for k, t in out.items():
        check(f"lora_oracle out[{k}]", oo[k].detach(), t)
    check("lora_oracle loss", lo_loss.detach(), loss.detach(), rtol=1e-6)
    for k, gr in grads.items():
        check(f"lora_oracle grad[{k}]", lo_grads[k], gr, rtol=1e-4, atol=1e-7)
    for k, (_, a, b) in adapters.items():
        stem = k.rsplit(".", 1)[0]
        check(f"lora_oracle dA[{stem}]", lo_grads[f"{LO.PEFT_PREFIX}{stem}.lora_A.siglip_adapter.weight"],
              scale * b.t() @ grad_w[k], rtol=1e-4, atol=1e-8)
        check(f"lora_oracle dB[{stem}]", lo_grads[f"{LO.PEFT_PREFIX}{stem}.lora_B.siglip_adapter.weight"],
              scale * grad_w[k] @ a.t(), rtol=1e-4, atol=1e-8)

    blob = {}
    for k, t in np_state(state).items():
        blob["state/" + k] = t
    for k, t in sample.items():
        blob["sample/" + k] = t.numpy()
    for k, t in out.items():
        blob["out/" + k] = t.numpy()
    blob["loss"] = loss.detach().numpy()
    for k, t in grads.items():
        blob["grad/" + k] = t.numpy()
    for k, t in grad_w.items():
        blob["gradw/" + k] = t.numpy()
    blob["meta/is_bimanual"] = np.array(int(is_bimanual))
    blob["meta/context_length"] = np.array(T if sequential else 0)
    blob["meta/r"], blob["meta/lora_alpha"] = np.array(r), np.array(lora_alpha)
    np.savez_compressed(os.path.join(GOLD, fname), **blob)
    print(f"  wrote {fname}")


def postproc_fixture(seed: int):
    """Heatmap -> action and pixel -> target-map vectors from the reference's own functions
    (models/utils.py:60-83, models/__init__.py:79-173, data/processor.py:405-439)."""
    print("postproc fixture")
    rng = np.random.default_rng(seed)
    size, batch = 24, 6
    out = {}

    def heat(n):
        h = rng.random((n, size, size), dtype=np.float32)
        h[0, 3, 5] = h[0, 17, 2] = 2.0            # tied maxima: the first in row-major order wins
        h[1] = 0.25                                # constant map -> pixel (0, 0)
        return h

    mask = np.zeros((batch, 1, size, size), dtype=np.float32)
    for i in range(batch):
        r0, c0 = rng.integers(0, size - 8, 2)
        mask[i, 0, r0:r0 + rng.integers(2, 8), c0:c0 + rng.integers(2, 8)] = 1.0
    mask[2, 0] = 0.0
    mask[2, 0, 4, 4] = mask[2, 0, 4, 20] = mask[2, 0, 20, 4] = 1.0   # equidistant candidates
    mask[3] = 1.0

    class Fixed(RefBaseModel):
        def __init__(self, outputs, **kw):
            super().__init__(image_size=size, device="cpu", **kw)
            self.outputs = outputs

        def forward(self, x):
            return self.outputs

    # unimanual, pick constrained to the mask
    uni = {"pick_heatmap": heat(batch), "place_heatmap": heat(batch)}
    uni["pick_heatmap"][2] = 0.0
    uni["pick_heatmap"][2, 12, 12] = 1.0          # (12,12): 3 mask pixels at distance^2 128 -> first
    sample = {"mask": torch.from_numpy(mask)}
    ref_act = Fixed({k: torch.from_numpy(v) for k, v in uni.items()}, is_bimanual=False).get_action(sample)
    mine = O.select_action(uni, mask, False)
    assert np.array_equal(ref_act.pick, mine["pick"]) and np.array_equal(ref_act.place, mine["place"])
    for k, v in uni.items():
        out[f"uni.{k}"] = v
    out["uni.expect.pick"], out["uni.expect.place"] = ref_act.pick, ref_act.place
    px, cf = ref_sample_from_heatmap(torch.from_numpy(uni["pick_heatmap"]), torch.from_numpy(mask),
                                     return_confidence=True)
    opx, ocf = O.sample_from_heatmap(uni["pick_heatmap"], mask)
    assert np.array_equal(px, opx) and np.array_equal(cf, ocf)
    out["uni.expect.pick_conf"] = cf

    # bimanual: confidences straddle the 0.01 threshold in every combination
    bi = {f"{a}_{k}_heatmap": heat(batch) * 0.5 for a in ("left", "right") for k in ("pick", "place")}
    scale = {"left": [1.0, 1.0, 0.001, 0.001, 0.004, 1.0], "right": [1.0, 0.001, 1.0, 0.002, 0.004, 1.0]}
    for a in ("left", "right"):
        bi[f"{a}_pick_heatmap"] *= np.asarray(scale[a], dtype=np.float32)[:, None, None]
    ref_act = Fixed({k: torch.from_numpy(v) for k, v in bi.items()}, is_bimanual=True).get_action(sample)
    mine = O.select_action(bi, mask, True)
    for k in ("left_pick", "right_pick", "left_place", "right_place"):
        assert np.array_equal(getattr(ref_act, k), mine[k]), k
        out[f"bi.expect.{k}"] = getattr(ref_act, k)
    for k, v in bi.items():
        out[f"bi.{k}"] = v
    out["mask"] = mask
    print("  [ok] get_action uni/bimanual: oracle == reference (exact)")

    # target maps
    gsize = 40
    cases = {"single": np.array([[10.5, 20.5]], np.float32),          # half-to-even rounding
             "single_b": np.array([[11.5, 3.49]], np.float32),
             "multi": np.array([[5.2, 6.7], [30.1, 30.9], [18.5, 2.5]], np.float32),
             "dup": np.array([[8.0, 8.0], [8.4, 7.6]], np.float32),
             "edge": np.array([[0.0, 39.0], [39.0, 0.0], [45.0, -3.0], [20.0, 20.0]], np.float32)}
    for strategy in ("first", "gmm"):
        proc = types.SimpleNamespace(image_size=gsize, cfg=types.SimpleNamespace(strategy=strategy, sigma=5))
        for name, pts in cases.items():
            ref = RefProcessor._make_gaussmap(proc, torch.from_numpy(pts)).numpy()
            mine = O.make_gaussmap(pts, gsize, 5, strategy)
            assert np.array_equal(ref, mine), (strategy, name, np.abs(ref - mine).max())
            out[f"gauss.{strategy}.{name}.points"] = pts
            out[f"gauss.{strategy}.{name}.expect"] = ref
    # `fit` (scipy multivariate_normal on the unrounded points): needs a non-singular covariance, i.e. >= 3 non-collinear
    # points; one point falls back to `first` like every strategy.  The closed-form restatement may differ from scipy's
    # eigen-decomposition in the last fp64 ulp, i.e. by at most one fp32 ulp after the cast.
    proc = types.SimpleNamespace(image_size=gsize, cfg=types.SimpleNamespace(strategy="fit", sigma=5))
    for name in ("single", "multi", "edge"):
        pts = cases[name]
        ref = RefProcessor._make_gaussmap(proc, torch.from_numpy(pts)).numpy()
        mine = O.make_gaussmap(pts, gsize, 5, "fit")
        assert np.allclose(ref, mine, rtol=2.4e-7, atol=1e-37), ("fit", name, np.abs(ref - mine).max())
        out[f"gauss.fit.{name}.points"] = pts
        out[f"gauss.fit.{name}.expect"] = ref
    ref = RefProcessor._make_gaussmap(proc).numpy()
    assert np.array_equal(ref, O.make_gaussmap(None, gsize)) and not ref.any()
    print("  [ok] _make_gaussmap first/gmm/None: oracle == reference (exact); fit: within 1 fp32 ulp")
    np.savez_compressed(os.path.join(GOLD, "postproc.npz"), **out)


def stats(t: torch.Tensor):
    t = t.detach().double().flatten()
    return np.concatenate([[t.mean().item(), t.std().item(), t.abs().max().item()],
                           t[:8].numpy()])


def full_size_summary(seed: int, batch: int):
    """Config A (ViT-B/16 @224, fusion 512/16 heads/8 layers, unimanual) at small batch:
    weights come from the ORACLE's seeded constructors (reproducible on any host
    from the seed alone), are loaded into the reference model, and the reference's
    loss / output / gradient summaries are frozen."""
    print(f"== full-size summary, config A, B={batch}")
    gen = torch.Generator().manual_seed(seed)
    state = {**O.make_clip_state(gen), **O.make_tail_state(gen)}
    clip_kwargs = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                       vision_patch_size=16, context_length=77, vocab_size=49408,
                       transformer_width=512, transformer_heads=8, transformer_layers=12)
    mk = dict(TINY_MODEL, image_size=224, depth=8, heads=16)
    model = build_ref_rgbonly(clip_kwargs, mk, seed)
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("attn_mask" in k or k == "" for k in missing), missing
    sample = O.synthetic_sample(batch, torch.Generator().manual_seed(seed + 11))
    rec = run_reference_training(model, sample, False, steps=1)
    with torch.no_grad():
        oo = O.rgb_clip_forward(state, sample, 16)
    blob = {"meta/seed": np.array(seed), "meta/batch": np.array(batch),
            "loss": rec["loss"].numpy()}
    for k, t in rec["out"].items():
        check(f"full out[{k}]", oo[k], t, rtol=1e-4, atol=1e-6)
        blob["out_stats/" + k] = stats(t)
        blob["out_head/" + k] = t[0, 100:108, 100:108].numpy()
    for k, t in rec["grads"].items():
        blob["grad_stats/" + k] = stats(t)
    np.savez_compressed(os.path.join(GOLD, "config_a_summary.npz"), **blob)
    print("  wrote config_a_summary.npz; loss =", float(rec["loss"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only-siglip", action="store_true")
    ap.add_argument("--only-lora", action="store_true")
    ap.add_argument("--only-postproc", action="store_true")
    ap.add_argument("--only-patch14", action="store_true")
    ap.add_argument("--only-mask-head", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if args.only_patch14:
        tiny_end_to_end(False, "tiny_patch14.npz", seed=3414, patch=14)
        return
    if args.only_mask_head:
        mask_head_fixtures()
        return
    if args.only_lora:
        siglip_lora_fixture(False, "tiny_siglip_lora.npz", seed=3415)
        siglip_lora_fixture(True, "tiny_siglip_sequential_lora.npz", seed=3416)
        return
    postproc_fixture(seed=3413)
    if args.only_postproc:
        return
    tiny_end_to_end(False, "tiny_patch14.npz", seed=3414, patch=14)
    siglip_fixture(False, "tiny_siglip.npz", seed=3411)
    siglip_fixture(True, "tiny_siglip_sequential.npz", seed=3412)
    siglip_lora_fixture(False, "tiny_siglip_lora.npz", seed=3415)
    siglip_lora_fixture(True, "tiny_siglip_sequential_lora.npz", seed=3416)
    if args.only_siglip:
        return
    tiny_end_to_end(False, "tiny_unimanual.npz", seed=3407)
    tiny_end_to_end(True, "tiny_bimanual.npz", seed=3408)
    masked_fusion_fixture(seed=3409)
    per_op_fixture(seed=3410)
    mask_head_fixtures()
    if not args.skip_full:
        full_size_summary(seed=3407, batch=2)


if __name__ == "__main__":
    main()

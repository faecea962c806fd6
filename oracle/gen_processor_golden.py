#!/usr/bin/env python3
"""Fixtures for the input side of the path, made by RUNNING THE REFERENCE (import from
/root/reference; this container only):

  tests/golden/tokenizer.npz   `clip.tokenize` ids (bifold/models/clip/__init__.py:241-287) of instructions
                               formatted from the reference's own templates (data/vr_folding_utils.py:13-66)
                               plus edge cases (case, punctuation, digits, apostrophes, repeated blanks)
  tests/golden/augment.npz     `Processor._aug_pixel` (data/processor.py:344-354) on annotated pixels for fixed
                               (angle, dx, dy), including an out-of-frame case (AssertionError -> valid = 0)
  tests/golden/depth.npz       the depth transform chain without its torchvision links (data/processor.py:59-78):
                               `DepthScale`, `MaskDepth`, float32, `TruncatedDepthStandardization`
                               (data/depth_augmentations.py:6-17, 35-51) on frames with tie runs at the cuts
                               (masked zeros), distinct values, and a constant frame

Stubs (none on the arithmetic): `ftfy` is absent -- `fix_text` is the identity on the ASCII templates used here;
`torchvision` / `open3d` / `plotly` / `trimesh` are import-time only.  The image half of the augmentation
(`v2.functional.affine`, `v2.Resize`) lives in torchvision, which is not installed: no fixture can be made for it
and the device kernels for it are checked against torch's own grid_sample / interpolate (parity unpinned).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_processor_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
GOLD = os.path.join(REPO, "tests", "golden")

for name in ("ftfy", "torchvision", "torchvision.transforms", "torchvision.transforms.v2", "open3d", "plotly",
             "plotly.graph_objects", "trimesh", "timm", "timm.models", "timm.models.vision_transformer"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
sys.modules["ftfy"].fix_text = getattr(sys.modules["ftfy"], "fix_text", lambda t: t)
tv = sys.modules["torchvision.transforms"]
for n in ("CenterCrop", "Compose", "Normalize", "Resize", "ToTensor", "InterpolationMode"):
    if not hasattr(tv, n):
        setattr(tv, n, type(n, (), {"BICUBIC": 3}))
if not hasattr(tv, "v2"):
    tv.v2 = sys.modules["torchvision.transforms.v2"]
if not hasattr(sys.modules["timm.models.vision_transformer"], "Block"):
    sys.modules["timm.models.vision_transformer"].Block = object

from bifold.models.clip import tokenize as ref_tokenize  # noqa: E402
from bifold.data.vr_folding_utils import folding_actions  # noqa: E402
from bifold.data.processor import Processor as RefProcessor  # noqa: E402


def tokenizer_fixture():
    texts = []
    for t in folding_actions["sleeves"][:6]:
        texts += [t.format(which="left"), t.format(which="right")]
    for t in folding_actions["refine"][:4]:
        texts.append(t.format(which="bottom", garment="tshirt"))
    for t in folding_actions["fold"][:8]:
        texts.append(t.format(garment="trousers", which1="left", which2="right"))
    texts += ["FOLD the T-Shirt's   left sleeve, twice!", "fold 2 times: top->bottom (50%)", "don't we'll it's i'm you've he'd",
              "a", "  leading and trailing blanks  ", "unfold"]
    ids = ref_tokenize(texts).numpy()
    assert ids.shape == (len(texts), 77) and ids.dtype == np.int32
    np.savez_compressed(os.path.join(GOLD, "tokenizer.npz"), texts=np.array(texts), ids=ids)
    print(f"wrote tokenizer.npz: {len(texts)} instructions, longest {int((ids != 0).sum(1).max())} tokens")


def augment_fixture():
    g = torch.Generator().manual_seed(3415)
    size = 223                                     # image_size - 1 (processor.py:318)
    cases, outs, valid = [], [], []
    params = [(-5.0, 3.0, -2.0), (4.5, -4.0, 5.0), (0.0, 0.0, 0.0), (2.25, 1.5, 0.5), (-3.0, 5.9, -5.9), (5.0, 5.0, 5.0)]
    for angle, dx, dy in params:
        for k in (1, 3):
            px = torch.rand(k, 2, generator=g) * 180 + 20
            if angle == 5.0:
                px[0] = torch.tensor([222.5, 0.2])          # leaves the frame -> the reference's assert fires
            try:
                out = RefProcessor._aug_pixel(None, px.clone(), angle, dx, dy, size)
                ok = 1
            except AssertionError:
                out, ok = torch.zeros_like(px), 0
            cases.append(np.concatenate([[angle, dx, dy, k], px.numpy().ravel(), np.zeros(6 - 2 * k)]))
            outs.append(np.concatenate([out.numpy().ravel(), np.zeros(6 - 2 * k)]))
            valid.append(ok)
    np.savez_compressed(os.path.join(GOLD, "augment.npz"), cases=np.array(cases, dtype=np.float64),
                        outs=np.array(outs, dtype=np.float32), valid=np.array(valid), size=np.array(size))
    print(f"wrote augment.npz: {len(cases)} cases, {sum(valid)} in frame")


def depth_fixture():
    """The reference's own classes, run in the order Processor.__init__ composes them (processor.py:59-78); the
    Resize / CenterCrop links in between are torchvision's (absent) and the identity at the frames' own size."""
    from bifold.data import depth_augmentations as DA
    S, B = 96, 6
    g = torch.Generator().manual_seed(3416)
    depth = (torch.rand(B, S, S, generator=g) * 0.6 + 0.9).numpy().astype(np.float32)      # metres from the camera
    mask = np.zeros((B, S, S), dtype=np.float32)
    for b, frac in enumerate((0.35, 0.05, 0.93, 0.5, 1.0, 0.0)):      # cloth coverage: cuts inside / outside the zero run
        k = int(round(frac * S))
        mask[b, :k, :] = 1.0
    mask[3] = (torch.rand(S, S, generator=g) > 0.5).float().numpy()
    depth[4] = 1.25                                                   # constant frame: variance 0
    depth[2].reshape(-1)[::7] = depth[2, 0, 0]                        # ties away from zero
    np.random.seed(3417)
    scale, masker, standardize = DA.DepthScale(-0.2, 0.2), DA.MaskDepth(), DA.TruncatedDepthStandardization()
    scaled, masked, out = [], [], []
    for b in range(B):
        sample = {"depth": depth[b].copy(), "mask": mask[b] if b != 5 else None}
        before = sample["depth"].copy()
        sample = scale(sample)
        assert sample["depth"].dtype == np.float32
        scaled.append(sample["depth"].copy())
        m = masker(sample)                                            # [S, S, 1]
        assert m.shape == (S, S, 1)
        t = torch.from_numpy(np.ascontiguousarray(m[:, :, 0])).to(torch.float32)     # v2.ToImage + ToDtype(float32)
        masked.append(t.numpy().copy())
        out.append(standardize(t).numpy())
    np.random.seed(3417)                                              # replay DepthScale's draws (one uniform per frame)
    shifts = [np.float32(np.random.uniform(-0.2, 0.2)) for _ in range(B)]
    for b in range(B):                                                # numpy adds the python float as a float32
        assert np.array_equal(scaled[b], depth[b] + shifts[b]), b
    np.savez_compressed(os.path.join(GOLD, "depth.npz"), depth=depth, mask=mask, has_mask=np.array([1, 1, 1, 1, 1, 0]),
                        shift=np.array(shifts, dtype=np.float32), masked=np.array(masked, dtype=np.float32),
                        standardized=np.array(out, dtype=np.float32), thresh=np.array(0.1))
    print(f"wrote depth.npz: {B} frames of {S}x{S}, shifts {np.round(shifts, 4).tolist()}")


if __name__ == "__main__":
    tokenizer_fixture()
    augment_fixture()
    depth_fixture()

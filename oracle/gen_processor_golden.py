#!/usr/bin/env python3
"""Fixtures for the input side of the path, made by RUNNING THE REFERENCE (import from
/root/reference; this container only):

  tests/golden/tokenizer.npz   `clip.tokenize` ids (bifold/models/clip/__init__.py:241-287) of instructions
                               formatted from the reference's own templates (data/vr_folding_utils.py:13-66)
                               plus edge cases (case, punctuation, digits, apostrophes, repeated blanks)
  tests/golden/augment.npz     `Processor._aug_pixel` (data/processor.py:344-354) on annotated pixels for fixed
                               (angle, dx, dy), including an out-of-frame case (AssertionError -> valid = 0)

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


if __name__ == "__main__":
    tokenizer_fixture()
    augment_fixture()

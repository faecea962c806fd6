"""The input side of the path: CLIP tokenisation, the Processor's rgb transform for frames
of any resolution, and its spatial augmentation.

Pinned by the reference itself (oracle/gen_processor_golden.py): `clip.tokenize` ids of instructions built from the
reference's templates (tests/golden/tokenizer.npz) and `Processor._aug_pixel` (tests/golden/augment.npz).
PARITY UNPINNED (torchvision is not in the image, so the reference's `v2.Resize` / `v2.functional.affine` cannot
be run): the device kernels are compared with oracle restatements over torch's own uint8 antialiased-bicubic
interpolate and grid_sample."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEAN, STD = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]


def test_clip_tokenizer_matches_reference_ids(monkeypatch):
    from bifold.models import clip_tokenizer as T
    if not os.environ.get("BIFOLD_CLIP_BPE") and not os.environ.get("BIFOLD_REFERENCE"):
        monkeypatch.setenv("BIFOLD_REFERENCE", "/root/reference")      # the build container's read-only checkout
    try:
        T.find_bpe_file()
    except FileNotFoundError:
        pytest.skip("CLIP's BPE merge table (a data file the reference vendors) is not on this machine")
    z = np.load(os.path.join(GOLD, "tokenizer.npz"))
    texts = [str(t) for t in z["texts"]]
    ids = T.tokenize(texts)
    assert ids.dtype == torch.int32 and tuple(ids.shape) == (len(texts), 77)
    assert np.array_equal(ids.numpy(), z["ids"])
    assert ids[0, 0] == 49406 and (ids == 49407).sum(1).eq(1).all()
    with pytest.raises(RuntimeError, match="too long"):
        T.tokenize("fold " * 100)
    cut = T.tokenize("fold " * 100, truncate=True)
    assert cut[0, -1] == 49407 and cut[0, 0] == 49406


def test_pixel_augmentation_matches_reference():
    from bifold.augment import aug_pixel, draw_augmentation
    z = np.load(os.path.join(GOLD, "augment.npz"))
    size = float(z["size"])
    for case, ref, ok in zip(z["cases"], z["outs"], z["valid"]):
        angle, dx, dy, k = case[0], case[1], case[2], int(case[3])
        px = torch.from_numpy(case[4:4 + 2 * k].reshape(k, 2)).float()
        got, inside = aug_pixel(px, angle, dx, dy, size)
        assert inside == bool(ok)
        if ok:
            assert torch.equal(got, torch.from_numpy(ref[:2 * k].reshape(k, 2)))
    # the retry loop: a pixel near the border forces redraws; the accepted draw keeps everything in frame
    rng = np.random.RandomState(0)
    pixels = {"pick": torch.tensor([[221.0, 110.0]]), "place": torch.tensor([[10.0, 3.0], [50.0, 60.0]]), "left_pick": None}
    moved, (angle, dx, dy) = draw_augmentation(pixels, 224, rng=rng)
    assert set(moved) == {"pick", "place"} and all((v >= 0).all() and (v < 223).all() for v in moved.values())
    assert -5 <= angle < 6 and -5 <= dx < 6 and -5 <= dy < 6
    never, p = draw_augmentation({"pick": torch.tensor([[222.9, 222.9]])}, 224, translate=(3, 6), rng=rng)
    assert never is None and p is None


def _camera_frames(h, w, n=2):
    g = torch.Generator().manual_seed(h * w)
    base = torch.rand(n, 3, h // 8 + 1, w // 8 + 1, generator=g)
    raw = (torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear") * 255).round().clamp(0, 255)
    raw = (raw + torch.randint(-12, 13, raw.shape, generator=g)).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    return raw, g


@pytest.mark.parametrize("h,w,size", [(480, 640, 224), (300, 300, 224), (224, 224, 224), (720, 540, 384), (100, 160, 224)])
def test_oracle_resize_is_witnessed_by_pil_bicubic(h, w, size):
    """torchvision is absent, so the oracle's `resize_normalize_rgb` cannot be pinned by the reference's own
    `rgb_transform`.  A second witness that IS here: PIL's antialiased BICUBIC, the routine the reference calls for
    `raw_rgb` (processor.py:263, `self.resize(Image.fromarray(rgb))`) and the one torchvision's uint8 tensor path is
    written to agree with.  Same sizes, same centre crop: at most one grey level apart when shrinking by 2x or more, two
    otherwise, on well under 1 % of the pixels (the two fixed-point schemes round ties differently)."""
    from PIL import Image
    from oracle import bifold_oracle as O
    raw, _ = _camera_frames(h, w, n=1)
    want = O.resize_normalize_rgb(raw, None, size, MEAN, STD)[0]
    new_h, new_w = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)
    img = np.asarray(Image.fromarray(raw[0].numpy()).resize((new_w, new_h), Image.Resampling.BICUBIC))
    top, left = int(round((new_h - size) / 2.0)), int(round((new_w - size) / 2.0))
    img = torch.from_numpy(img[top:top + size, left:left + size].astype(np.float32)).permute(2, 0, 1) / 255.0
    pil = (img - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
    lsb = (pil - want).abs() * torch.tensor(STD).view(3, 1, 1) * 255.0
    assert float(lsb.max()) <= (1.01 if min(h, w) >= 2 * size or min(h, w) == size else 2.01)
    assert float((lsb > 0.5).float().mean()) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,size", [(480, 640, 224), (300, 300, 224), (224, 224, 224), (720, 540, 384), (100, 160, 224)])
def test_resize_normalize_rgb_matches_torch_uint8_bicubic(gpu, observed, h, w, size):
    from oracle import bifold_oracle as O
    from bifold_amd.hip import ops
    g = torch.Generator().manual_seed(h * w)
    base = torch.rand(2, 3, h // 8 + 1, w // 8 + 1, generator=g)
    raw = (torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear") * 255).round().clamp(0, 255)
    raw = (raw + torch.randint(-12, 13, raw.shape, generator=g)).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    mask = (torch.rand(2, h, w, generator=g) > 0.2).float()
    for m in (None, mask):
        want = O.resize_normalize_rgb(raw, m, size, MEAN, STD)
        got = ops.resize_normalize_rgb(raw.to(gpu), None if m is None else m.to(gpu), size, MEAN, STD).cpu()
        assert got.shape == want.shape == (2, 3, size, size)
        std = torch.tensor(STD).view(1, 3, 1, 1)
        lsb = ((got - want).abs() * std * 255.0)                 # difference in uint8 steps of the resized image
        observed(f"resize_{h}x{w}_to_{size}_{'mask' if m is not None else 'nomask'}", max_lsb=float(lsb.max()),
                 frac_off=float((lsb > 0.5).float().mean()))
        # downsampling (camera frames are larger than the model's input): never more than one grey level, and only
        # where a value sits on a rounding edge (observed 0.35 % of the pixels); upsampling, where torch's uint8
        # kernel works with 2^-7-quantised integer weights: two levels at most
        assert float(lsb.max()) <= (1.01 if min(h, w) >= size else 2.01)
        assert float((lsb > 0.5).float().mean()) < 0.02


@pytest.mark.gpu
def test_affine_nearest_matches_grid_sample(gpu):
    from oracle import bifold_oracle as O
    from bifold_amd.hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 64, 64, generator=g)
    params = torch.tensor([[0.0, 0.0, 0.0], [5.0, -3.5, 2.25], [-4.2, 5.9, -5.9], [1.0, 0.0, 4.0]])
    got = ops.affine_nearest(x.to(gpu), params.to(gpu)).cpu()
    for i in range(4):
        want = O.affine_nearest(x[i:i + 1], *[float(v) for v in params[i]])
        diff = (got[i:i + 1] != want).float().mean().item()
        assert diff < 2e-3, (i, diff)                            # only pixels whose source lands on a .5 tie may differ
    assert torch.equal(got[0], x[0])
    assert (got[1] == 0).any()                                   # zero fill where the source leaves the frame


@pytest.mark.gpu
def test_augment_batch_moves_images_and_annotations_together(gpu):
    """A bright dot at the annotated pixel must follow its annotation through the augmentation's TRANSLATION (same
    (dx, dy) for `_aug_pixel` and `v2.functional.affine`, processor.py:313-341).  For the rotation the reference
    hands `-angle` to the pixels and `angle` to torchvision's affine; both are reproduced as written (the pixel half
    pinned by tests/golden/augment.npz, the image half against the oracle's restatement of torchvision)."""
    from oracle import bifold_oracle as O
    from bifold.augment import augment_batch
    S, B, T = 96, 6, 2
    rng = np.random.RandomState(3)
    pick = torch.tensor(rng.uniform(25, 70, (B, 2)), dtype=torch.float32).round()
    rgb = torch.zeros(B, 3, S, S)
    ctx = torch.zeros(B, T, 3, S, S)
    for b in range(B):
        x, y = int(pick[b, 0]), int(pick[b, 1])
        rgb[b, :, y - 1:y + 2, x - 1:x + 2] = 1.0
        ctx[b, :, :, y - 1:y + 2, x - 1:x + 2] = 1.0
    sample = {"rgb": rgb.to(gpu), "rgb_context": ctx.to(gpu), "pick": pick.clone(), "pick_heatmap": torch.zeros(B, S, S),
              "instruction": torch.zeros(B, 77, dtype=torch.int32)}
    shift_only = dict(rotate_augmentation=[0, 0], translate_augmentation=[-5, 6], max_augmentation_trials=5)
    out = augment_batch(sample, S, shift_only, rng=rng)
    assert out["pick_heatmap"] is sample["pick_heatmap"] and out["augmentation"].shape == (B, 3)
    assert not torch.equal(out["pick"], pick)
    for b in range(B):
        img = out["rgb"][b, 0].cpu()
        ys, xs = torch.nonzero(img > 0.5, as_tuple=True)
        cx, cy = xs.float().mean().item(), ys.float().mean().item()
        assert abs(cx - out["pick"][b, 0].item()) < 0.51 and abs(cy - out["pick"][b, 1].item()) < 0.51
        assert torch.equal(out["rgb_context"][b, 0], out["rgb"][b]) and torch.equal(out["rgb_context"][b, 1], out["rgb"][b])
    # with rotation: images follow the oracle's affine for the drawn parameters, pixels follow _aug_pixel(-angle)
    full = dict(rotate_augmentation=[-5, 6], translate_augmentation=[-5, 6], max_augmentation_trials=5)
    out = augment_batch(sample, S, full, rng=rng)
    from bifold.augment import aug_pixel
    for b in range(B):
        angle, dx, dy = [float(v) for v in out["augmentation"][b].cpu()]
        want = O.affine_nearest(rgb[b:b + 1], angle, dx, dy)
        assert (out["rgb"][b:b + 1].cpu() != want).float().mean().item() < 2e-3
        moved, inside = aug_pixel(pick[b:b + 1].clone(), -angle, dx, dy, S - 1)
        assert inside and torch.allclose(out["pick"][b], moved[0], atol=1e-4)


# ------------------------------------------------------------------ depth frames
def _depth_fixture():
    return np.load(os.path.join(GOLD, "depth.npz"))


def test_oracle_depth_chain_matches_reference_run():
    """tests/golden/depth.npz = the reference's DepthScale / MaskDepth / TruncatedDepthStandardization run on six frames
    (tie runs at the cuts, ties away from zero, a constant frame, no mask); the oracle's restatement reproduces it."""
    from oracle import bifold_oracle as O
    z = _depth_fixture()
    mask = np.where(z["has_mask"][:, None, None] > 0, z["mask"], 1.0).astype(np.float32)
    got_m = O.depth_frames(z["depth"], mask, z["shift"])
    assert np.array_equal(got_m, z["masked"])
    got = O.depth_frames(z["depth"], mask, z["shift"], standardize=True, thresh=float(z["thresh"]))
    assert np.allclose(got, z["standardized"], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_depth_frames_kernels_match_reference_run_and_oracle(gpu, observed):
    """bf_depth_prepare + bf_depth_standardize (radix select of the two cut values, tie counts, fp64 trimmed moments)
    against the reference-run fixture, then against the oracle on 224-px frames: masked cloth (a long run of exact
    zeros through the lower cut), random masks, constant frames."""
    from bifold_amd.hip import ops
    from oracle import bifold_oracle as O
    z = _depth_fixture()
    mask = np.where(z["has_mask"][:, None, None] > 0, z["mask"], 1.0).astype(np.float32)
    d, m, sh = (torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in (z["depth"], mask, z["shift"]))
    assert torch.equal(ops.depth_frames(d, m, sh).cpu(), torch.from_numpy(z["masked"]))
    got = ops.depth_frames(d, m, sh, standardize=True, thresh=float(z["thresh"])).cpu().numpy()
    err = np.abs(got - z["standardized"]).max() / np.abs(z["standardized"]).max()
    assert err < 2e-6, err
    g = torch.Generator().manual_seed(41)
    B, S = 8, 224
    depth = torch.rand(B, 1, S, S, generator=g) + 0.8
    mask = torch.zeros(B, 1, S, S)
    for b in range(B):
        mask[b, 0, : 20 + 25 * b] = 1.0
    mask[5] = (torch.rand(1, S, S, generator=g) > 0.3).float()
    depth[6] = 2.0
    mask[6] = 1.0
    shift = torch.rand(B, generator=g) * 0.4 - 0.2
    ref = O.depth_frames(depth[:, 0].numpy(), mask[:, 0].numpy(), shift.numpy(), standardize=True)
    got = ops.depth_frames(depth.to(gpu), mask.to(gpu), shift.to(gpu), standardize=True)
    assert got.shape == depth.shape
    err2 = np.abs(got[:, 0].cpu().numpy() - ref).max() / np.abs(ref).max()
    observed("depth_frames", fixture=float(err), oracle_224=float(err2))
    assert err2 < 2e-6, err2

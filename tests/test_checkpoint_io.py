"""Pretrained-weight files a reference user already has load into the towers:
OpenAI's TorchScript CLIP archives (bifold/models/clip/__init__.py:135-142), Hugging Face
safetensors snapshots (bifold/models/siglip.py:26), plain and nested state dicts.  CPU only."""
import os
import warnings

import pytest
import torch
from torch import nn

from bifold.models import checkpoint_io as cio
from bifold.models.clip_encoder import CLIPEncoder
from bifold.models.params import add_param
from bifold.models.siglip_encoder import SiglipTowers

TINY_CLIP = dict(embed_dim=32, image_resolution=32, vision_layers=2, vision_width=64, vision_patch_size=8,
                 context_length=77, vocab_size=512, transformer_width=32, transformer_heads=2,
                 transformer_layers=2)
TINY_SIGLIP = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2)


def _randomise(module: nn.Module, seed: int):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    return {k: v.clone() for k, v in module.state_dict().items()}


def _same(module: nn.Module, want):
    got = module.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_clip_from_torchscript_archive_and_plain_state_dicts(tmp_path):
    src = CLIPEncoder(**TINY_CLIP)
    want = _randomise(src, 1)
    # an archive shaped like OpenAI's: a scripted module whose state dict has the CLIP keys plus
    # the three bookkeeping tensors build_model deletes (clip/model.py:550-552)
    shell = nn.Module()
    for k, v in want.items():
        add_param(shell, k, v.clone(), requires_grad=False)
    for k, v in (("input_resolution", 32), ("context_length", 77), ("vocab_size", 512)):
        shell.register_buffer(k, torch.tensor(v))
    jit_path = str(tmp_path / "ViT-tiny.pt")
    torch.jit.save(torch.jit.script(shell), jit_path)
    sd = cio.read_state_dict(jit_path)
    assert "input_resolution" not in sd and set(sd) == set(want)
    dst = CLIPEncoder(**TINY_CLIP)
    cio.load_into(dst, sd, "CLIP tiny")
    _same(dst, want)

    plain, nested = str(tmp_path / "plain.pt"), str(tmp_path / "nested.pth")
    torch.save({k: v.half() for k, v in want.items()}, plain)           # fp16, as the published weights are
    torch.save({"epoch": 3, "model": want}, nested)
    for path, exact in ((plain, False), (nested, True)):
        dst = CLIPEncoder(**TINY_CLIP)
        cio.load_into(dst, cio.read_state_dict(path), "CLIP tiny")
        got = dst.state_dict()
        for k in want:
            assert got[k].dtype == torch.float32
            assert torch.equal(got[k], want[k] if exact else want[k].half().float())


def test_siglip_from_safetensors_file_and_snapshot_directory(tmp_path):
    from safetensors.torch import save_file
    vis = dict(TINY_SIGLIP, image_size=32, patch_size=16)
    txt = dict(TINY_SIGLIP, vocab_size=100, max_position_embeddings=16)
    want = _randomise(SiglipTowers(vis, txt), 2)
    snap = tmp_path / "snapshot"
    snap.mkdir()
    save_file({k: v.contiguous() for k, v in want.items()}, str(snap / "model.safetensors"))
    for path in (str(snap / "model.safetensors"), str(snap)):
        dst = SiglipTowers(vis, txt)
        cio.load_into(dst, cio.read_state_dict(path), "SigLIP tiny")
        _same(dst, want)


def test_incomplete_or_mismatched_checkpoints_are_refused(tmp_path):
    src = CLIPEncoder(**TINY_CLIP)
    want = _randomise(src, 3)
    short = {k: v for k, v in want.items() if "ln_final" not in k}
    with pytest.raises(KeyError, match="lacks 2 of"):
        cio.load_into(CLIPEncoder(**TINY_CLIP), short, "CLIP tiny")
    wide = CLIPEncoder(**dict(TINY_CLIP, transformer_width=64))
    with pytest.raises(ValueError, match="wrong shape"):
        cio.load_into(wide, want, "CLIP wide")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cio.load_into(CLIPEncoder(**TINY_CLIP), dict(want, **{"visual.extra": torch.zeros(1)}), "CLIP tiny")
    assert any("ignoring 1 checkpoint entries" in str(x.message) for x in w)
    with pytest.raises(FileNotFoundError):
        cio.read_state_dict(str(tmp_path / "absent.pt"))
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        cio.read_state_dict(str(tmp_path / "empty"))
    torch.save([1, 2, 3], str(tmp_path / "list.pt"))
    with pytest.raises(ValueError, match="not a state dict"):
        cio.read_state_dict(str(tmp_path / "list.pt"))


def test_tower_inside_a_full_model_checkpoint():
    src = CLIPEncoder(**TINY_CLIP)
    want = _randomise(src, 4)
    full = {"clip_encoder." + k: v for k, v in want.items()}
    full["project.weight"] = torch.zeros(4, 4)
    dst = CLIPEncoder(**TINY_CLIP)
    cio.load_into(dst, full, "CLIP tiny", prefix="clip_encoder.")
    _same(dst, want)

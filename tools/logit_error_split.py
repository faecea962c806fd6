#!/usr/bin/env python3
"""Which part of the bf16 step carries the logit error against the fp32 CPU oracle at random initialisation?

    python3 tools/logit_error_split.py [n_samples=4] [batch=16]

Runs config A's forward with (towers, tail) in (bf16, bf16), (fp32, bf16), (bf16, fp32), (fp32, fp32) and prints the
heatmap / logit errors of each against oracle/bifold_oracle.py on the same weights and inputs (diagnostic only).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bifold.models.pickplace import RGBTailFn  # noqa: E402


def forward_split(model, sample, tower_dt, tail_dt):
    rgb, instr = sample["rgb"], sample["instruction"]
    enc = model.clip_encoder
    enc._runtime(tower_dt)
    text = enc.encode_text_with_embeddings(instr, tower_dt)                       # fp32 out
    vis = enc.encode_image_with_embeddings(rgb, tower_dt, out_dtype=tail_dt)      # ln_post written in the tail's dtype
    model.compute_dtype = tail_dt
    model.pick_place.fusion.compute_dtype = tail_dt
    model._train_arena = None
    arena = model.trainable_arena()
    arena.sync_shadow()
    model.pick_place.collapse_async()
    heats = RGBTailFn.apply(model, text, vis, *[p for _, p in model.trainable_params()])
    return {f"{n}_heatmap": h for n, h in zip(model.pick_place.heads_out, heats)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    from oracle import bifold_oracle as O
    dev = torch.device("cuda:0")
    import warnings
    from bifold.models import Models
    torch.manual_seed(3407)
    cfg = dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
               patch_size=16, depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="bf16")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = Models.get_by_name(cfg, device=dev).to(dev)
    sample = bench.synthetic_batch(batch, dev, seed=3407)
    state = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    sub = {k: v[:n].detach().cpu() for k, v in sample.items() if isinstance(v, torch.Tensor)}
    with torch.no_grad():
        ref = O.rgb_clip_forward(state, sub, heads=model.pick_place.fusion.heads)
    model.eval()
    for tower_dt, tail_dt in ((torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16),
                              (torch.bfloat16, torch.float32), (torch.float32, torch.float32)):
        with torch.no_grad():
            out = forward_split(model, sample, tower_dt, tail_dt)
        hr = lr = 0.0
        for k in ("pick_heatmap", "place_heatmap"):
            got, want = out[k][:n].double().cpu(), ref[k].double()
            hr = max(hr, ((got - want).abs().max() / want.abs().max()).item())
            lg, lw = torch.logit(got.clamp(1e-300, 1 - 1e-16)), torch.logit(want.clamp(1e-300, 1 - 1e-16))
            lr = max(lr, ((lg - lw).abs().max() / lw.abs().max()).item())
        print(f"towers {str(tower_dt)[6:]:>8}  tail {str(tail_dt)[6:]:>8}:  heatmap max rel {hr:.3e}   logit max rel {lr:.3e}", flush=True)


if __name__ == "__main__":
    main()

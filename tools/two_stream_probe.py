#!/usr/bin/env python3
"""Upper bound of running the step as TWO half-batches on two HIP streams (tails of one stream's persistent
GEMMs filled by the other stream's kernels): two independent model replicas at B/2 each, steps issued back to
back from one host thread, against one replica at B.  Diagnostic only (tools/README.md)."""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from bifold.losses import Losses
from bifold.models import Models
from bifold.optim import Optimizers

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = 10


def make(batch, seed):
    cfg = dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
               patch_size=16, depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="bf16")
    torch.manual_seed(3407)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Models.get_by_name(cfg, device=dev).to(dev).train()
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
    opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                                 params=[p for p in m.parameters() if p.requires_grad])
    sample = bench.synthetic_batch(batch, dev, seed)

    def step():
        out = m(sample)
        loss, _ = loss_fn(out, sample)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    return step


def timed(fn, steps=STEPS, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


one = make(B, 1)
t1 = timed(one)
print(f"one stream, B={B}: {1e3 * t1:.2f} ms/step  {B / t1:.0f} samples/s", flush=True)
del one
torch.cuda.empty_cache()
for parts in (2, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    steps = []
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            steps.append(make(B // parts, 10 + i))

    def both():
        for s, st in zip(streams, steps):
            with torch.cuda.stream(s):
                st()
    t2 = timed(both)
    print(f"{parts} streams, B={B // parts} each: {1e3 * t2:.2f} ms per {B} samples  {B / t2:.0f} samples/s", flush=True)
    del steps, both
    torch.cuda.empty_cache()

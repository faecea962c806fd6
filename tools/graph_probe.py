#!/usr/bin/env python3
"""Experiment: capture forward + loss + backward of the config-A step into a HIP graph
(torch.cuda.graph) and replay it; the optimiser step stays eager (its step count / LR are host
scalars).  Prints eager vs replayed ms/step."""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bifold.losses import Losses  # noqa: E402
from bifold.models import Models  # noqa: E402
from bifold.optim import Optimizers  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(3407)
cfg = dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
           patch_size=16, depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
           pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype=os.environ.get("PROBE_DTYPE", "fp16"))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = Models.get_by_name(cfg, device=dev).to(dev)
model.train()
loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                             params=[p for p in model.parameters() if p.requires_grad])
sample = bench.synthetic_batch(256, dev, seed=3407)


def fwd_bwd():
    out = model(sample)
    loss, _ = loss_fn(out, sample)
    opt.zero_grad()
    loss.backward()
    return loss


def timeit(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager():
    loss = fwd_bwd()
    opt.step()
    return loss


for _ in range(3):
    eager()
print(f"eager: {timeit(eager):.2f} ms/step", flush=True)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = fwd_bwd()


def replay():
    g.replay()
    opt.step()


for _ in range(3):
    replay()
print(f"graph replay + eager Adam: {timeit(replay):.2f} ms/step; loss {static_loss.item():.5f}", flush=True)
l0 = static_loss.item()
for _ in range(20):
    replay()
print(f"loss after 20 more replayed steps: {static_loss.item():.5f} (was {l0:.5f})")

#!/usr/bin/env python3
"""Where do the small device-to-device copies of a training step come from?  One step of config A under torch.profiler
with Python stacks; prints the call sites of every aten::copy_ / hipMemcpyAsync / fill of the step (diagnostic)."""
import os, sys, warnings, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bifold.models import Models
from bifold.losses import Losses
from bifold.optim import Optimizers

dev = torch.device("cuda:0")
torch.manual_seed(3407)
cfg = dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
           patch_size=16, depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
           pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="bf16")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = Models.get_by_name(cfg, device=dev).to(dev)
model.train()
loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                             params=[p for p in model.parameters() if p.requires_grad])
sample = bench.synthetic_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 64, dev, seed=3407)


def step():
    out = model(sample)
    loss, _ = loss_fn(out, sample)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "/root/repo" in s or "bifold" in s]
        sites[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, where), n in sites.most_common(40):
    print(f"{n:4d}  {name:18s} {where}")
names = collections.Counter(ev.name for ev in prof.events() if "Memcpy" in ev.name or "Memset" in ev.name or "copyBuffer" in ev.name or "fillBuffer" in ev.name)
print(dict(names))

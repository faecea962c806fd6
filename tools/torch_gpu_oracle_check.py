#!/usr/bin/env python3
"""Does the CPU oracle, evaluated by torch's own fp32 kernels on the GPU, reproduce its CPU gradients?  (diagnostic)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bifold_oracle as O

def run(state, sample, dev, trainable):
    st = {k: (v.to(dev).clone().requires_grad_(True) if k in trainable else v.to(dev)) for k, v in state.items()}
    sm = {k: v.to(dev) for k, v in sample.items()}
    out = O.rgb_clip_forward(st, sm, heads=16)
    loss, _ = O.bce_gaussmap(out, sm, False)
    grads = torch.autograd.grad(loss, [st[k] for k in trainable])
    return loss.item(), {k: g.detach().double().cpu() for k, g in zip(trainable, grads)}, {k: v.detach().double().cpu() for k, v in out.items() if k.endswith("heatmap")}

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    gen = torch.Generator().manual_seed(3407)
    state = {**O.make_clip_state(gen), **O.make_tail_state(gen)}
    sample = O.synthetic_sample(B, torch.Generator().manual_seed(3418))
    trainable = sorted(k for k in state if O.is_trainable_key(k))
    t0 = time.time(); lc, gc, hc = run(state, sample, torch.device("cpu"), trainable); print("cpu", time.time() - t0, lc, flush=True)
    for flag in (None, "highest"):
        if flag:
            torch.set_float32_matmul_precision(flag)
            torch.backends.cuda.matmul.allow_tf32 = False
            torch.backends.cudnn.allow_tf32 = False
        lg, gg, hg = run(state, sample, torch.device("cuda:0"), trainable)
        print("flag", flag, "loss rel", abs(lg - lc) / abs(lc), "heat", max(((hg[k] - hc[k]).abs().max() / hc[k].abs().max()).item() for k in hc))
        errs = {k: ((gg[k] - gc[k]).abs().max() / gc[k].abs().max()).item() for k in trainable}
        ratio = {k: (gg[k].norm() / gc[k].norm()).item() for k in trainable}
        worst = sorted(errs, key=errs.get)[-5:]
        for k in worst:
            print("   ", k, "rel", errs[k], "norm ratio", ratio[k])
        print("   median rel", sorted(errs.values())[len(errs) // 2], flush=True)

if __name__ == "__main__":
    main()

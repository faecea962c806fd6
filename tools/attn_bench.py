#!/usr/bin/env python3
"""Attention forward / backward microbenchmark on the shapes of the BiFold step (B = 256)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops  # noqa: E402

SHAPES = [("vision", 256, 197, 12, 64, 0), ("text", 256, 77, 8, 64, 1), ("fusion", 256, 275, 16, 32, 0)]


SIGLIP_SHAPES = [("vis224", 64, 196, 12, 64, 0), ("txt64", 64, 64, 12, 64, 0), ("fus224", 32, 262, 16, 48, 0), ("fus300", 32, 300, 12, 64, 0), ("vision384", 64, 576, 12, 64, 0), ("text", 16, 64, 12, 64, 0), ("fusion", 16, 2373, 16, 48, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--only", default="", help="comma-separated shape names")
    ap.add_argument("--family", default="clip", choices=["clip", "siglip"],
                    help="siglip: the long sequences of siglip_sequential at 384 px (B = 16, T = 3)")
    ap.add_argument("--shape", action="append", default=[],
                    help="name,B,L,H,D,mode -- a custom shape (repeatable; replaces the family's list)")
    args = ap.parse_args()
    shapes = SHAPES if args.family == "clip" else SIGLIP_SHAPES
    if args.shape:
        shapes = [(f[0],) + tuple(int(x) for x in f[1:]) for f in (sh.split(",") for sh in args.shape)]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    hdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    for name, B, L, H, D, mode in shapes:
        if args.only and name not in args.only.split(","):
            continue
        qkv = torch.randn(B * L, 3 * H * D, device=dev, generator=g).to(hdt)
        dout = torch.randn(B * L, H * D, device=dev, generator=g).to(hdt)
        rm = None
        if mode == 2:       # siglip_sequential's mask: [text | T context frames of 577 rows | current frame], the
            rm = torch.ones(B, L, dtype=torch.int32, device=dev)      # first k ~ U{0..T} context frames valid
            t = max((L - 65 - 577) // 577, 0) if L > 1300 else max((L - 65 - 197) // 197, 0)
            k = torch.randint(0, t + 1, (B,), generator=torch.Generator().manual_seed(1))
            for bi in range(B):
                fr = 577 if L > 1300 else 197
                rm[bi, 65 + int(k[bi]) * fr: 65 + t * fr] = 0
        out, lse = ops.attention_fwd(qkv, B, L, H, D, D ** -0.5, mode, rm, need_lse=True)
        fns = {"fwd": lambda: ops.attention_fwd(qkv, B, L, H, D, D ** -0.5, mode, rm, need_lse=True)}
        if args.bwd:
            fns["bwd"] = lambda: ops.attention_bwd(qkv, out, dout, lse, B, L, H, D, D ** -0.5, mode, rm)
        for tag, fn in fns.items():
            ts = []
            for r in range(args.rounds + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1) / 3)
            ts.sort()
            fl = (4.0 if tag == "fwd" else 10.0) * B * H * L * L * D * (0.5 if mode == 1 else 1.0)
            print(f"{name:7s} {tag} B{B} L{L} H{H} D{D} mask{mode}: {ts[len(ts)//2]*1e3:7.1f} us  {fl/ts[len(ts)//2]/1e9:6.1f} TF/s")


if __name__ == "__main__":
    main()

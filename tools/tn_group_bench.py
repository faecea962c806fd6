#!/usr/bin/env python3
"""The grouped weight-gradient launch (bf_gemm_tn_grouped: the four dW = dY^T X of one fusion layer) over a range of token
counts K: config A's 70 400 rows up to siglip_sequential's 377 600 (B = 128 x 2 950).  Same 48 output tiles, same five
K splits -- only the length of a split's walk (and the footprint all CUs stream at once) changes.

    python tools/tn_group_bench.py [--dtype fp16] [--rows 70400,141312,188800,377600] [--rounds 5]

Prints per K: time, TF/s, unique operand GB/s, and microseconds per 64-row K tile of one work item.
"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--rows", default="70400,141312,188800,377600")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--random", type=int, default=1, help="0: zero operands (no data-dependent power)")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    D, M = args.dim, 4 * args.dim
    for k in [int(r) for r in args.rows.split(",")]:
        mk = (lambda *s: torch.randn(*s, device=dev, dtype=dt)) if args.random else (lambda *s: torch.zeros(*s, device=dev, dtype=dt))
        h1, dqkv = mk(k, D), mk(k, 3 * D)
        o, dx2 = mk(k, D), mk(k, D)
        h2, da = mk(k, D), mk(k, M)
        a, dx3 = mk(k, M), mk(k, D)
        outs = [torch.empty(3 * D, D, device=dev), torch.empty(D, D, device=dev), torch.empty(M, D, device=dev),
                torch.empty(D, M, device=dev)]
        probs = [(dqkv, h1, outs[0]), (dx2, o, outs[1]), (da, h2, outs[2]), (dx3, a, outs[3])]
        if not ops.gemm_tn_grouped(probs):
            print(f"K = {k}: the group does not qualify")
            continue
        torch.cuda.synchronize()
        times = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_tn_grouped(probs)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(times)[len(times) // 2]
        flops = 2.0 * k * (3 * D * D + D * D + 2 * M * D)
        ubytes = 2.0 * k * (4 * D + 3 * D + 2 * M + D + D)        # every operand once
        tiles_k = (k / 5) / 64
        ref = (dx2[:4096].float().t() @ o[:4096].float())
        print(f"K = {k:7d}: {t:9.1f} us  {flops / t / 1e6:7.1f} TF/s  unique operands {ubytes / t / 1e3:7.1f} GB/s  "
              f"{t / tiles_k:6.2f} us per K tile of an item   (min {min(times):.1f}, max {max(times):.1f})")
        del ref


if __name__ == "__main__":
    main()

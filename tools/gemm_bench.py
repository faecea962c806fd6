#!/usr/bin/env python3
"""Per-shape GEMM microbenchmark on the shapes of the BiFold training step (B = 256).
Interleaved rounds in one process (guide §5.4 rule 24), random bf16 operands (rule 25)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops  # noqa: E402

SHAPES = [  # name, layout, m, n, k
    ("vis qkv", "nt", 50432, 2304, 768), ("vis out", "nt", 50432, 768, 768),
    ("vis fc1", "nt", 50432, 3072, 768), ("vis fc2", "nt", 50432, 768, 3072),
    ("txt qkv", "nt", 19712, 1536, 512), ("txt fc1", "nt", 19712, 2048, 512),
    ("fus qkv", "nt", 70400, 1536, 512), ("fus out", "nt", 70400, 512, 512),
    ("fus fc1", "nt", 70400, 2048, 512), ("fus fc2", "nt", 70400, 512, 2048),
    ("fus dA", "nn", 70400, 2048, 512), ("fus dH", "nn", 70400, 512, 2048), ("fus dqkv", "nn", 70400, 512, 1536),
    ("big 8k", "nt", 8192, 8192, 8192), ("big 4k", "nt", 4096, 4096, 4096),
    ("fus dW2", "tn", 512, 2048, 70400), ("fus dW1", "tn", 2048, 512, 70400), ("fus dWqkv", "tn", 1536, 512, 70400),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--variants", default="-1", help="comma list of bf_gemm_set_variant values to interleave")
    ap.add_argument("--split", type=int, default=0, help="force split_k (0 = the library default)")
    ap.add_argument("--check", action="store_true", help="compare every variant with torch.matmul first")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--blaslt", action="store_true", help="also time hipBLASLt (torch.matmul) on the same buffers")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    from bifold_amd.hip import lib as L
    dev = torch.device("cuda:0")
    hdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g = torch.Generator(device=dev).manual_seed(0)
    cases = []
    for name, lay, m, n, k in SHAPES:
        if args.only and args.only not in name:
            continue
        ta, bkn = lay[0] == "t", lay[1] == "n"
        a = torch.randn((k, m) if ta else (m, k), device=dev, generator=g).to(hdt)
        b = (torch.randn((k, n) if bkn else (n, k), device=dev, generator=g) * k ** -0.5).to(hdt)
        out = torch.empty(m, n, device=dev, dtype=torch.float32 if ta else hdt)
        cases.append((name, lay, m, n, k, a, b, out, ta, bkn))
    if args.check:
        for name, lay, m, n, k, a, b, out, ta, bkn in cases:
            ref = ((a.float().t() if ta else a.float()) @ (b.float() if bkn else b.float().t()))
            for v in variants:
                L.lib.bf_gemm_set_variant(v)
                out.zero_()
                ops.gemm(a, b, trans_a=ta, b_kn=bkn, out=out, split_k=(args.split or None))
                err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                print(f"check {name:10s} variant {v:2d}: rel err {err:.2e} {'ok' if err < 1e-2 else 'FAIL'}", flush=True)
            del ref
    if args.blaslt:
        variants = variants + [99]
    res = {(c[0], v): [] for c in cases for v in variants}
    for r in range(args.rounds + 1):
        for name, lay, m, n, k, a, b, out, ta, bkn in cases:
            for v in (variants[r % len(variants):] + variants[:r % len(variants)]):   # rotate: no variant always runs on the other's warm caches
                if v == 99:      # the vendor library on the same operands (a ceiling estimate, not product code)
                    if ta:
                        continue
                    A_, B_ = a, (b if bkn else b.t())
                    o_ = torch.empty(m, n, device=dev, dtype=hdt)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        torch.matmul(A_, B_, out=o_)
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0:
                        res[(name, v)].append(e0.elapsed_time(e1) / 3)
                    continue
                L.lib.bf_gemm_set_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.gemm(a, b, trans_a=ta, b_kn=bkn, out=out, split_k=(args.split or None))
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    res[(name, v)].append(e0.elapsed_time(e1) / 3)
    for v in variants:
        tot_ms, tot_fl = 0.0, 0.0
        for name, lay, m, n, k, *_ in cases:
            ts = sorted(res[(name, v)])
            if not ts:
                continue
            med = ts[len(ts) // 2]
            fl = 2.0 * m * n * k
            tot_ms += med
            tot_fl += fl
            print(f"v{v:2d} {name:10s} {lay} {m:6d}x{n:5d}x{k:6d}  {med * 1e3:8.1f} us  {fl / med / 1e9:7.1f} TF/s (min {fl / ts[0] / 1e9:7.1f})")
        print(f"v{v:2d} TOTAL {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TF/s")
    L.lib.bf_gemm_set_variant(-1)


if __name__ == "__main__":
    main()

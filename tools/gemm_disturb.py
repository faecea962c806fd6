#!/usr/bin/env python3
"""How does the persistent GEMM behave when some CUs are held by another stream's resident kernel
(what RCCL's collectives do during backward)?  A spin kernel takes N CUs on a side stream for ~2 ms
while the GEMM is timed on the main stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
for (m, n, k) in [(50432, 2304, 768), (70400, 2048, 512)]:
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    b = torch.randn(n, k, device=dev, generator=g).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for variant in (8, 4):
        L.lib.bf_gemm_set_variant(variant)
        for taken in (0, 8, 32):
            ts = []
            for rep in range(6):
                torch.cuda.synchronize()
                if taken:
                    L.check(L.lib.bf_debug_spin(taken, 4_000_000, side.cuda_stream), "spin")   # ~2 ms
                    torch.cuda._sleep(200_000)          # let the spin blocks become resident first
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(a, b, out=out)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(f"{m}x{n}x{k} variant {variant} CUs taken {taken:2d}: {ts[len(ts)//2]:7.1f} us", flush=True)
L.lib.bf_gemm_set_variant(-1)

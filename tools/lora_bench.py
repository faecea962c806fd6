#!/usr/bin/env python3
"""The unmerged LoRA adapter branch of one layer at configs[2]'s shapes (vision tower: 128 x 5 frames x 576 tokens = 368 640
rows, width 768, q_proj + v_proj, r = 8, lora_dropout 0.01): csrc/lora.hip's kernels, each with the bytes it has to move,
beside the dropout + GEMM form they replace.  usage: lora_bench.py [--rows N] [--dtype fp16|bf16] [--iters K]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bifold_amd.hip import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=368640)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--p", type=float, default=0.01)
    a = ap.parse_args()
    cdt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    M, W, r, p = a.rows, a.width, 8, a.p
    h1 = torch.randn(M, W, device=dev).to(cdt)
    qkv = torch.randn(M, 3 * W, device=dev).to(cdt)
    dh1 = torch.randn(M, W, device=dev).to(cdt)
    A = (torch.randn(2 * r, W, device=dev) / 28).to(cdt)
    Bt = (torch.randn(2 * r, W, device=dev) / 28).to(cdt)
    u = torch.randn(M, 2 * r, device=dev)
    seeds, offs, zero = [5, 9], [0, 2 * W], [0, 0]
    gA = [torch.empty(r, W, device=dev) for _ in range(2)]
    gB = [torch.empty(W, r, device=dev) for _ in range(2)]
    wide = M * W * 2 / 1e6                                  # MB of one [rows, width] 16-bit tensor
    rows = [
        ("fwd  down  u = (D . h1) A^T          ", lambda: ops.lora_down(h1, zero, A, r, W, p, seeds, alpha=1.01), wide),
        ("fwd  up    qkv[:, q|v] += u Bt       ", lambda: ops.lora_up(u, Bt, qkv, offs, r, W), 4 * wide),
        ("bwd  down  du = dqkv[:, q|v] Bt^T    ", lambda: ops.lora_down(qkv, offs, Bt, r, W), 2 * wide),
        ("bwd  tn    dB = dqkv[:, q|v]^T u     ", lambda: ops.lora_tn(qkv, offs, u, gB, r, W, True, alpha=4.0), 2 * wide),
        ("bwd  tn    dA = du^T (D . h1)        ", lambda: ops.lora_tn(h1, zero, u, gA, r, W, False, p, seeds, alpha=1.01), wide),
        ("bwd  up    dh1 += D . (du A)         ", lambda: ops.lora_up(u, A, dh1, zero, r, W, p, seeds, alpha=1.01), 2 * wide),
        ("bwd  both  the two rows above, one walk ", lambda: ops.lora_bwd_input(h1, u, A, dh1, gA, r, W, p, seeds, alpha=1.01), 3 * wide),
    ]
    total = 0.0
    print(f"# rows {M} width {W} r {r} slots 2 {a.dtype} p {p}")
    for i, (name, fn, mb) in enumerate(rows):
        us = timed(fn, a.iters)
        total += us if i not in (4, 5) else 0.0                   # the backward runs the fused row instead of rows 4 + 5
        print(f"{name} {us:8.1f} us   {mb:7.0f} MB   {mb / us:6.2f} TB/s")
    print(f"fused form, one layer forward + backward: {total:8.1f} us")
    # the form of rounds 3 - 5: per projection dropout -> N = r GEMM -> K = r GEMM (+ the backward's five passes)
    a16, b16 = A[:r].contiguous(), Bt[:r].t().contiguous()
    u16 = u[:, :r].to(cdt).contiguous()

    def old_fwd():
        for s in range(2):
            hd = ops.dropout(h1, p, seeds[s])
            uu = ops.gemm(hd, a16)
            ops.gemm(uu, b16, alpha=4.0, out=qkv[:, offs[s]:offs[s] + W], accumulate=True)

    def old_bwd():
        for s in range(2):
            dy = qkv[:, offs[s]:offs[s] + W]
            du = ops.gemm(dy, b16, b_kn=True, alpha=4.0)
            ops.gemm(dy, u16, trans_a=True, b_kn=True, alpha=4.0, out=gB[s])
            hd = ops.dropout(h1, p, seeds[s])
            ops.gemm(du, hd, trans_a=True, b_kn=True, out=gA[s])
            t = ops.gemm(du, a16, b_kn=True)
            ops.dropout(t, p, seeds[s], out=dh1, accumulate=True)

    # the forward's up-projection as a low-rank addend of the packed projection's GEMM instead of the 'fwd up' pass
    wq = (torch.randn(3 * W, W, device=dev) / 28).to(cdt)
    bq = torch.randn(3 * W, device=dev)
    uh = u.to(cdt)
    bcat = torch.zeros(3 * W, 2 * r, device=dev, dtype=cdt)
    bcat[:W, :r] = Bt[:r].t()
    bcat[2 * W:, r:] = Bt[r:].t()
    if ops.gemm_lowrank_ok(M, 3 * W, W, cdt, 2 * r):
        g0 = timed(lambda: ops.gemm(h1, wq, bias=bq), a.iters)
        g1 = timed(lambda: ops.gemm(h1, wq, bias=bq, lr_u=uh, lr_w=bcat), a.iters)
        g0b = timed(lambda: ops.gemm(h1, wq, bias=bq), a.iters)
        print(f"packed projection GEMM {M} x {3 * W} x {W}: {g0:8.1f} us plain, {g1:8.1f} us with the low-rank addend, {g0b:8.1f} us plain again")
    f, b = timed(old_fwd, max(2, a.iters // 4)), timed(old_bwd, max(2, a.iters // 4))
    print(f"dropout + GEMM form: forward {f:8.1f} us, backward {b:8.1f} us, together {f + b:8.1f} us")


if __name__ == "__main__":
    main()

"""LayerNorm forward timing at the step's shapes (fp32 in -> bf16 / fp32 out).

    BIFOLD_LN_ROWS=1|2|4 python tools/ln_bench.py

Prints us per call and the algorithmic GB/s (read x + write y).  The activations of the
B = 256 step are rotated through several buffers so the input is not simply resident in L2.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bifold_amd.hip import ops

SHAPES = [(256 * 197, 768, torch.bfloat16), (256 * 77, 512, torch.bfloat16),
          (256 * 274, 512, torch.bfloat16), (256 * 274, 512, torch.float32)]


def main():
    dev = torch.device("cuda:0")
    for rows, cols, odt in SHAPES:
        xs = [torch.randn(rows, cols, device=dev) for _ in range(4)]
        g = torch.randn(cols, device=dev); b = torch.randn(cols, device=dev)
        outs = [torch.empty(rows, cols, dtype=odt, device=dev) for _ in range(4)]
        ref = torch.nn.functional.layer_norm(xs[0], (cols,), g, b, 1e-5)
        y, mu, rs = ops.layernorm_fwd(xs[0], g, b, odt, save_stats=True)
        err = (y.float() - ref).abs().max().item()
        for i in range(8):
            ops.layernorm_fwd(xs[i % 4], g, b, odt, out=outs[i % 4])
        torch.cuda.synchronize()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            ops.layernorm_fwd(xs[i % 4], g, b, odt, out=outs[i % 4])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        nbytes = rows * cols * (4 + (2 if odt == torch.bfloat16 else 4))
        print(f"rows={rows:6d} cols={cols:4d} out={str(odt)[6:]:8s} {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s  max|err|={err:.3g}")


if __name__ == "__main__":
    main()

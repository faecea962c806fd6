"""The residual add + LayerNorm passes of the frozen towers at the step's shapes (fp16 branches), isolated:
   rmw   x += d ; y = LN(x)                 12 bytes per element   (the two passes per block until round 5)
   ro    y = LN(x + d)                       8                     (after the attention branch: no x_out)
   two   x = (x + d) + d2 ; y = LN(x)       14                     (after the MLP: both branches at once)
Prints us per call and the algorithmic TB/s.  Buffers are rotated so that the inputs are not resident in L2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bifold_amd.hip import ops

SHAPES = [("vision", 256 * 197, 768), ("text", 256 * 77, 512), ("fusion", 256 * 275, 512)]


def timed(fn, n=100):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    cdt = torch.float16
    for name, rows, cols in SHAPES:
        xs = [torch.randn(rows, cols, device=dev) for _ in range(4)]
        ds = [torch.randn(rows, cols, device=dev).to(cdt) for _ in range(4)]
        d2 = [torch.randn(rows, cols, device=dev).to(cdt) for _ in range(4)]
        g = torch.randn(cols, device=dev); b = torch.randn(cols, device=dev)
        n_el = rows * cols
        forms = [("rmw", 12, lambda i: ops.add_layernorm_fwd(xs[i % 4], ds[i % 4], g, b, cdt)),
                 ("ro", 8, lambda i: ops.add_layernorm_fwd(xs[i % 4], ds[i % 4], g, b, cdt, write_x=False)),
                 ("two", 14, lambda i: ops.add_layernorm_fwd(xs[i % 4], ds[i % 4], g, b, cdt, delta2=d2[i % 4]))]
        line = [f"{name:7s} {rows:6d} x {cols:4d}"]
        us = {}
        for tag, bpe, fn in forms:
            us[tag] = timed(fn)
            line.append(f"{tag} {us[tag]:6.1f} us {n_el * bpe / us[tag] / 1e6:5.2f} TB/s")
        line.append(f"| per block: rmw + rmw {2 * us['rmw']:6.1f} us -> ro + two {us['ro'] + us['two']:6.1f} us")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()

"""Phase timeline of attn_bwd_fused32_kernel (a -DATTN_PROBE build through BIFOLD_HIP_LIB).  Stamps of wave 0 per workgroup
(100 MHz wall clock): 0 entry, 1 images + row constants in LDS, 4 last query block done; 2 / 3: summed ticks in the key
blocks / waiting at the two barriers per query block."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, Lq, H, D = 256, 275, 16, 32
qkv = torch.randn(B * Lq, 3 * H * D, device=dev, generator=g).half()
dout = torch.randn(B * Lq, H * D, device=dev, generator=g).half()
out, lse = ops.attention_fwd(qkv, B, Lq, H, D, D ** -0.5, 0, None, need_lse=True)
for _ in range(3):
    ops.attention_bwd(qkv, out, dout, lse, B, Lq, H, D, D ** -0.5, 0, None)
torch.cuda.synchronize()
fn = L.lib.bf_debug_attn_probe
fn.restype = ctypes.c_int
n = 16384
buf = (ctypes.c_ulonglong * (n * 8))()
fn(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)[:B * H]
us = lambda x: x / 100.0
t0 = a[:, 0].min()
print(f"workgroups {len(a)}; kernel span {us(a[:, 4].max() - t0):.1f} us")
for nm, v in (("load phase (0->1)", a[:, 1] - a[:, 0]), ("nine query blocks (1->4)", a[:, 4] - a[:, 1]),
              ("  of which in key blocks + partial stores", a[:, 2]), ("  of which at the barriers", a[:, 3]),
              ("whole workgroup (0->4)", a[:, 4] - a[:, 0])):
    print(f"  {nm:44s} mean {us(v.mean()):7.2f} us  p10 {us(np.percentile(v, 10)):7.2f}  p50 {us(np.percentile(v, 50)):7.2f}  p90 {us(np.percentile(v, 90)):7.2f}")
life = (a[:, 4] - a[:, 0]).sum()
print(f"  average workgroups alive: {life / max(a[:, 4].max() - t0, 1):.1f} (= {life / max(a[:, 4].max() - t0, 1) / 256:.2f} per CU)")
print(f"  per 32 x 32 block and wave (27 blocks): {us((a[:, 2]).mean()) / 27 * 1e3:.0f} ns in compute")

"""Phase timeline of attn_bwd_onepass_kernel (a -DATTN_PROBE build through BIFOLD_HIP_LIB) on the sequential fusion's shape.
One wave per role stamps a 100 MHz wall clock: 0 entry, 4 exit (S wave 0); 1 / 2 / 6: S wave 0's summed ticks in its blocks / at
the step barriers / in stash + fetch (the wait for the next query block's rows); 3: dK wave 8 in its products (stash + fetch
included); 5: dQ wave 10 likewise."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, Lq, H, D = 16, 2373, 16, 48
qkv = torch.randn(B * Lq, 3 * H * D, device=dev, generator=g).half()
dout = torch.randn(B * Lq, H * D, device=dev, generator=g).half()
rm = torch.ones(B, Lq, dtype=torch.int32, device=dev)
out, lse = ops.attention_fwd(qkv, B, Lq, H, D, D ** -0.5, 2, rm, need_lse=True)
for _ in range(2):
    ops.attention_bwd(qkv, out, dout, lse, B, Lq, H, D, D ** -0.5, 2, rm)
torch.cuda.synchronize()
fn = L.lib.bf_debug_attn_probe
fn.restype = ctypes.c_int
n = 16384
buf = (ctypes.c_ulonglong * (n * 8))()
fn(buf)
nq = (Lq + 31) // 32
nwg = B * H * ((nq + 7) // 8)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)[:nwg]
us = lambda x: x / 100.0
t0 = a[:, 0].min()
print(f"workgroups {nwg}, {nq} steps each; kernel span {us(a[:, 4].max() - t0):.1f} us")
for nm, v in (("whole workgroup (0->4)", a[:, 4] - a[:, 0]), ("S wave 0: in its key block", a[:, 1]), ("S wave 0: at the step barriers", a[:, 2]),
              ("S wave 0: stash + fetch", a[:, 6]), ("dK wave 8: stash + fetch + products", a[:, 3]), ("dQ wave 10: stash + fetch + sum + store", a[:, 5])):
    print(f"  {nm:44s} mean {us(v.mean()):8.2f} us  p10 {us(np.percentile(v, 10)):8.2f}  p90 {us(np.percentile(v, 90)):8.2f}   per step {us(v.mean()) / nq * 1e3:6.0f} ns")

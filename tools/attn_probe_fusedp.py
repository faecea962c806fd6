"""Phase timeline of attn_bwd_fused32p_kernel (a -DATTN_PROBE build through BIFOLD_HIP_LIB; BIFOLD_ATTN_BWD_FUSED_P=2).  One wave
per role and workgroup stamps a 100 MHz wall clock: 0 entry, 4 exit (key wave 0); 2 / 3 / 7: key wave 0's summed ticks in its
steps / at the step barriers / between items; 5 / 6: wave 9's ticks in its dQ sums inside the steps / after the last step;
1: loader wave 10's ticks from its first request to its last deposit (barriers included), 10: of which at the barriers; 8: the
summing wave at the step barriers; 11 / 12: key wave 8 in its steps / at the barriers.  Sixteen slots per workgroup."""
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
ncu = torch.cuda.get_device_properties(0).multi_processor_count
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[:ncu * 16].reshape(ncu, 16)      # 16 slots per workgroup
items = B * H / ncu
us = lambda x: x / 100.0
t0 = a[:, 0].min()
print(f"workgroups {len(a)}, {items:.1f} items each; kernel span {us(a[:, 4].max() - t0):.1f} us")
for nm, v in (("whole workgroup (0->4)", a[:, 4] - a[:, 0]),
              ("key wave 0: in its steps", a[:, 2]), ("key wave 0: at the step barriers", a[:, 3]), ("key wave 0: between items", a[:, 7]),
              ("wave 9: dQ sums inside the steps", a[:, 5]), ("wave 9: the last dQ sum, after the last step", a[:, 6]),
              ("wave 9: at the step barriers", a[:, 8]),
              ("key wave 8: in its steps", a[:, 11]), ("key wave 8: at the step barriers", a[:, 12]),
              ("loader wave 10: its nine steps incl. barriers", a[:, 1]), ("loader wave 10: at the step barriers", a[:, 10])):
    print(f"  {nm:40s} mean {us(v.mean()):8.2f} us  p10 {us(np.percentile(v, 10)):8.2f}  p90 {us(np.percentile(v, 90)):8.2f}   per item {us(v.mean()) / items:6.2f} us  per step {us(v.mean()) / items / 9 * 1e3:6.0f} ns")

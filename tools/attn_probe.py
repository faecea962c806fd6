"""Phase timeline of attn_bwd_q_kernel (build: make with -DATTN_PROBE -> BIFOLD_HIP_LIB).  Stamps per workgroup
(100 MHz wall clock): 0 entry, 1 global loads issued + LDS written, 2 past the workgroup barrier, 3 key loop done,
4 stores complete."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, Lq, H, D = 256, 275, 16, 32
qkv = torch.randn(B * Lq, 3 * H * D, device=dev, generator=g).bfloat16()
dout = torch.randn(B * Lq, H * D, device=dev, generator=g).bfloat16()
out, lse = ops.attention_fwd(qkv, B, Lq, H, D, D ** -0.5, 0, None, need_lse=True)
for _ in range(3):
    ops.attention_bwd(qkv, out, dout, lse, B, Lq, H, D, D ** -0.5, 0, None)
torch.cuda.synchronize()
fn = L.lib.bf_debug_attn_probe
fn.restype = ctypes.c_int
n = 16384
buf = (ctypes.c_ulonglong * (n * 8))()
fn(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
wgs = int(os.environ.get("PROBE_SPLIT", "2")) * B * H
a = a[:min(wgs, n)]
t0 = a[:, 0].min()
us = lambda x: x / 100.0
print(f"workgroups {len(a)}; kernel span {us(a[:,4].max() - t0):.1f} us")
names = ["load issue + LDS write (0->1)", "barrier wait (1->2)", "key loop (2->3)", "dQ stores drained (3->4)", "whole workgroup (0->4)"]
segs = [a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3], a[:, 4] - a[:, 0]]
for nm, sgm in zip(names, segs):
    print(f"  {nm:34s} mean {us(sgm.mean()):7.2f} us  p10 {us(np.percentile(sgm,10)):7.2f}  p50 {us(np.percentile(sgm,50)):7.2f}  p90 {us(np.percentile(sgm,90)):7.2f}")
# slots 5-7: wave 0's summed ticks in the per-block prologues (until the operand loads have arrived) and key loops, blocks
nb = np.maximum(a[:, 7], 1)
print(f"  wave 0, per 32-query block: prologue (q / dO / O loads arrived) {us((a[:, 5] / nb).mean()):.2f} us, "
      f"nine key steps {us((a[:, 6] / nb).mean()):.2f} us = {us((a[:, 6] / nb).mean()) / 9 * 1e3:.0f} ns per step; blocks per wave {a[:, 7].mean():.2f}")
# concurrency: average number of workgroups alive
life = (a[:, 4] - a[:, 0]).sum()
print(f"  average workgroups alive: {life / max(a[:,4].max() - t0, 1):.1f}  (= {life / max(a[:,4].max() - t0, 1) / 256:.2f} per CU)")
ev = np.concatenate([np.stack([a[:, 0], np.ones(len(a), dtype=np.int64)], 1), np.stack([a[:, 4], -np.ones(len(a), dtype=np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
alive = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0], append=ev[-1, 0])
order = np.argsort(alive)
cw = np.cumsum(dt[order]) / max(dt.sum(), 1)
q = lambda f: int(alive[order][np.searchsorted(cw, f)])
print(f"  workgroups alive, time-weighted: p10 {q(0.1)}  p50 {q(0.5)}  p90 {q(0.9)}  max {int(alive.max())}  (256 CUs)")
starts = np.sort(a[:, 0] - t0)
print("  start time of the 1st / 768th / 1536th / last workgroup (us):", us(starts[0]), us(starts[min(767, len(starts)-1)]), us(starts[min(1535, len(starts)-1)]), us(starts[-1]))

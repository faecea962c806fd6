"""Cycles per stage of the persistent GEMM by operand layout (does a 64-byte K row cost more DMA time than a 256-byte one?)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fn = L.lib.bf_debug_p8_probe
fn.restype = ctypes.c_int
L.lib.bf_gemm_set_variant(8)
m, n, k = 50432, 2304, 768
for name, ta, bkn in [("nt", False, False), ("nn", False, True), ("tn", True, True), ("tt", True, False)]:
    a = torch.randn((k, m) if ta else (m, k), device=dev, generator=g).bfloat16()
    b = (torch.randn((k, n) if bkn else (n, k), device=dev, generator=g) * k ** -0.5).bfloat16()
    f = lambda: ops.gemm(a, b, trans_a=ta, b_kn=bkn, split_k=1)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf)
    print(f"{name}: {ms*1e3:7.1f} us {2*m*n*k/ms/1e9:5.0f} TF/s | main {buf[0]} cyc = {buf[0]/max(buf[2],1):.0f}/stage, clock {buf[0]/max(buf[1],1)*100:.0f} MHz, epilogue {buf[3]-buf[5]}, variant {L.lib.bf_gemm_last_variant()}", flush=True)

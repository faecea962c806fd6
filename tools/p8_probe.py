import ctypes, sys, os, torch
sys.path.insert(0, "/root/repo")
from bifold_amd.hip import ops, lib as L
L.lib.bf_gemm_set_variant(int(os.environ.get("PROBE_VARIANT", "8")))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fn = L.lib.bf_debug_p8_probe
fn.restype = ctypes.c_int
for (m, n, k, res) in [(50432, 2304, 768, False), (50432, 768, 3072, False), (50432, 768, 768, True), (50432, 768, 3072, True)]:
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    b = torch.randn(n, k, device=dev, generator=g).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.float32 if res else torch.bfloat16)
    kw = dict(bias=torch.randn(n, device=dev, generator=g), residual=torch.randn(m, n, device=dev, generator=g), split_k=1) if res else {}
    for _ in range(3):
        ops.gemm(a, b, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(a, b, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf)
    cyc, wall, nk, ecyc, ewall = buf[0], buf[1], buf[2], buf[3], buf[4]
    print(f"{m}x{n}x{k} res={res}: {ms*1e3:.1f} us {2*m*n*k/ms/1e9:.0f} TF/s | main loop {cyc} cyc / {nk} stages = {cyc/max(nk,1):.0f} cyc/stage (ideal 1024), wall ticks {wall} -> clock {cyc/max(wall,1)*100:.0f} MHz ; decode+prologue issue {buf[5]} cyc, epilogue {ecyc - buf[5]} cyc, next-item vmcnt(0)+barrier {buf[6]} cyc")

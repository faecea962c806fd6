"""In-kernel cycle probe of the persistent GEMM on the step's forward shapes (build: tools/build_probe.sh, run with
BIFOLD_HIP_LIB=bifold_amd/lib/libbifold_hip_probe.so).  Per tile of workgroup 0: main loop, prologue issue of the next
item, epilogue, and the vmcnt(0)+barrier wait at the start of the next item."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
variant = int(sys.argv[1]) if len(sys.argv) > 1 else -1      # 8 = BK 32 loop, 9 = BK 64 full-line loop, -1 = the library's choice
hdt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.bfloat16
L.lib.bf_gemm_set_variant(variant)
print(f"# variant {variant}, {hdt}")
g = torch.Generator(device=dev).manual_seed(0)
fn = L.lib.bf_debug_p8_probe
fn.restype = ctypes.c_int
shapes = [
    ("vis qkv", 50432, 2304, 768, dict(bias=1)),
    ("vis fc1", 50432, 3072, 768, dict(bias=1, act=L.ACT_QUICK_GELU)),
    ("vis out", 50432, 768, 768, dict(bias=1)),
    ("vis fc2", 50432, 768, 3072, dict(bias=1)),
    ("fus qkv", 70400, 1536, 512, dict()),
    ("fus fc1+g'", 70400, 2048, 512, dict(bias=1, act=L.ACT_ERF_GELU_SAVEGRAD, want_pre=True)),
    ("fus fc1", 70400, 2048, 512, dict(bias=1, act=L.ACT_ERF_GELU)),
    ("fus out+res", 70400, 512, 512, dict(bias=1, res=1)),
    ("fus fc2+res", 70400, 512, 2048, dict(bias=1, res=1)),
    ("big 8k", 8192, 8192, 8192, dict()),
]
for name, m, n, k, o in shapes:
    a = torch.randn(m, k, device=dev, generator=g).to(hdt)
    b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).to(hdt)
    kw = {}
    if o.get("bias"):
        kw["bias"] = torch.randn(n, device=dev, generator=g)
    if o.get("act"):
        kw["act"] = o["act"]
    if o.get("want_pre"):
        kw["want_pre"] = True
    if o.get("res"):
        kw["residual"] = torch.randn(m, n, device=dev, generator=g)
        kw["out_dtype"] = torch.float32
    for _ in range(3):
        ops.gemm(a, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(a, b, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf)
    clk = buf[0] / max(buf[1], 1) * 100
    sec = (ctypes.c_ulonglong * 8)()
    if hasattr(L.lib, "bf_debug_p8_probe_sec"):
        L.lib.bf_debug_p8_probe_sec(sec)
    nt64 = max(k // 64, 1)
    secs = " | per K tile, load0/mfma0/load1/mfma1: w0 " + "/".join(f"{sec[i] / nt64:.0f}" for i in range(4)) + \
           " w4 " + "/".join(f"{sec[4 + i] / nt64:.0f}" for i in range(4))
    per64 = buf[0] / max(k // 64, 1)
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    per_tile = buf[0] + buf[3] + buf[6]
    print(f"{name:12s} {m}x{n}x{k}: {ms*1e3:7.1f} us {2*m*n*k/ms/1e9:5.0f} TF/s | tiles {tiles} = {tiles/256:.2f} rounds | main {buf[0]:6d} cyc "
          f"({per64:5.0f} per 64 of K, ideal 2048) clock {clk:4.0f} MHz | prologue issue {buf[5]:5d} | epilogue {buf[3]-buf[5]:6d} | "
          f"next-item wait {buf[6]:6d} | epi halves (cvt+ds_write / ds_read+store) {buf[8]}/{buf[9]} {buf[10]}/{buf[11]} | sum/tile {per_tile} cyc = {per_tile/clk:.1f} us -> x rounds {per_tile/clk*tiles/256:.0f} us" + secs, flush=True)

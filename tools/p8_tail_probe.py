"""In-kernel stamps of the K chunks of a split tail (build: tools/build_probe.sh; BIFOLD_HIP_LIB=...probe.so): for the
first 64 workgroups that ran a chunk — main loop, accumulator dump + drain + barrier, arrival add, read-back (last arriver
only), main-loop-end to item-end; chunk index, chunks, arrival order."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops, lib as L
dev = torch.device("cuda:0")
L.lib.bf_gemm_set_variant(9)
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("vis fc2", 50432, 768, 3072), ("fus fc2", 70400, 512, 2048), ("fus dqkv-like nt", 70400, 512, 1536)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev, generator=g).half()
    b = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).half()
    bias = torch.randn(n, device=dev, generator=g)
    for _ in range(3): ops.gemm(a, b, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(a, b, bias=bias)
    e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    L.lib.bf_debug_p8_probe_tail(buf)
    main = (ctypes.c_ulonglong * 16)(); L.lib.bf_debug_p8_probe(main)
    print(f"{name} {m}x{n}x{k}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us; whole tile of workgroup 0: main {main[0]} cyc, rest {main[3]} cyc")
    rows = [[buf[w * 8 + i] for i in range(8)] for w in range(64)]
    rows = [(w, r) for w, r in enumerate(rows) if r[6] > 0]
    for w, r in rows[:12]:
        print(f"   wg {w:2d}: chunk {r[5]}/{r[6]} arrived #{r[7]} | main {r[0]:6d} | dump+drain {r[1]:6d} | add {r[2]:5d} | read-back {r[3]:6d} | loop end -> item end {r[4]:6d}")
    if rows:
        import statistics as st
        last = [r for _, r in rows if r[7] == r[6] - 1]
        notlast = [r for _, r in rows if r[7] != r[6] - 1]
        if last: print(f"   last arrivers ({len(last)}): read-back median {st.median(r[3] for r in last):.0f}, loop end -> item end median {st.median(r[4] for r in last):.0f}")
        if notlast: print(f"   others ({len(notlast)}): dump+drain median {st.median(r[1] for r in notlast):.0f}, loop end -> item end median {st.median(r[4] for r in notlast):.0f}")

#!/usr/bin/env python3
"""Per-SHAPE fabric traffic of the step's GEMMs (the in-step PMC summary only gives launch-weighted means per kernel
instantiation).  Two modes:

  run        (under `rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE` and again `--pmc WRITE_SIZE`)
             launches every shape of tools/gemm_bench.SHAPES with the step's epilogue REPS times, each case behind a
             pair of marker kernels (bf_cast of 2^20 / 2^21 elements) so the dispatches can be attributed
  summarize  <fetch_dir> <write_dir>  ->  table: measured read (x2 gfx950 correction) / write MB per launch against
             the algorithmic bytes (A + B once, residual / aux once, every output once)
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 4

# name, layout, m, n, k, epilogue
CASES = [
    ("vis qkv", "nt", 50432, 2304, 768, "bias"), ("vis out", "nt", 50432, 768, 768, "res"),
    ("vis fc1", "nt", 50432, 3072, 768, "act"), ("vis fc2", "nt", 50432, 768, 3072, "res"),
    ("fus qkv", "nt", 70400, 1536, 512, ""), ("fus out", "nt", 70400, 512, 512, "res"),
    ("fus fc1", "nt", 70400, 2048, 512, "pre"), ("fus fc2", "nt", 70400, 512, 2048, "res"),
    ("fus dA", "nn", 70400, 2048, 512, "grad"), ("fus dH", "nn", 70400, 512, 2048, ""),
    ("fus dqkv", "nn", 70400, 512, 1536, ""), ("fus do", "nn", 70400, 512, 512, ""),
    ("fus dW2", "tn", 512, 2048, 70400, ""), ("fus dW1", "tn", 2048, 512, 70400, ""),
    ("fus dWqkv", "tn", 1536, 512, 70400, ""), ("fus dWo", "tn", 512, 512, 70400, ""),
]


def algorithmic(lay, m, n, k, epi):
    ta = lay[0] == "t"
    rd = (m * k + n * k) * 2
    wr = m * n * (4 if (ta or epi == "res") else 2)
    if epi == "res":
        rd += m * n * 4
    if epi == "grad":
        rd += m * n * 2
    if epi == "pre":
        wr += m * n * 2
    return rd, wr


def run():
    import torch
    from bifold_amd.hip import lib as L
    from bifold_amd.hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    marker = torch.zeros(1 << 21, device=dev)
    mout = torch.empty_like(marker, dtype=torch.bfloat16)
    for idx, (name, lay, m, n, k, epi) in enumerate(CASES):
        ta, bkn = lay[0] == "t", lay[1] == "n"
        a = torch.randn((k, m) if ta else (m, k), device=dev, generator=g).bfloat16()
        b = torch.randn((k, n) if bkn else (n, k), device=dev, generator=g).bfloat16()
        kw = {}
        if epi in ("bias", "res", "act", "pre"):
            kw["bias"] = torch.randn(n, device=dev, generator=g)
        if epi == "res":
            kw["residual"] = torch.randn(m, n, device=dev, generator=g)
            kw["out_dtype"] = torch.float32
        if epi == "act":
            kw["act"] = L.ACT_QUICK_GELU
        if epi == "pre":
            kw["act"], kw["want_pre"] = L.ACT_ERF_GELU, True
        if epi == "grad":
            kw["act"], kw["aux"] = L.ACT_ERF_GELU_GRAD, torch.randn(m, n, device=dev, generator=g).bfloat16()
        if ta:
            kw["out"] = torch.empty(m, n, device=dev, dtype=torch.float32)
        ops.gemm(a, b, trans_a=ta, b_kn=bkn, **kw)              # warm (code objects, LDS attribute)
        torch.cuda.synchronize()
        ops.cast(marker[:1 << 20], torch.bfloat16, out=mout[:1 << 20])       # START marker
        for _ in range(REPS):
            ops.gemm(a, b, trans_a=ta, b_kn=bkn, **kw)
        ops.cast(marker, torch.bfloat16, out=mout)                           # END marker
        torch.cuda.synchronize()
        del a, b, kw


def load(d, counter):
    """Dispatches between a START marker (cast of 2^20 elements) and the next END marker (2^21) form one case."""
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    grids = sorted({int(r["Grid_Size"]) for r in rows if "cast_kernel" in r["Kernel_Name"]})
    assert len(grids) == 2, f"expected two marker sizes, found grids {grids}"
    out, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "cast_kernel" in name:
            if int(r["Grid_Size"]) == grids[0]:
                cur = {"kib": 0.0, "launches": 0, "kernels": {}}
            elif cur is not None:
                out.append(cur)
                cur = None
            continue
        if cur is None:
            continue
        cur["kib"] += float(r["Counter_Value"])
        cur["launches"] += 1
        key = name.split("(")[0][:60]
        cur["kernels"][key] = cur["kernels"].get(key, 0) + 1
    return out


def summarize(fd, wd):
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print(f"{'case':10s} {'layout':6s} {'m x n x k':>20s} {'epi':5s} {'read MB':>9} {'alg read':>9} {'x':>5} "
          f"{'write MB':>9} {'alg write':>9} {'x':>5}  kernels per call")
    for (name, lay, m, n, k, epi), f, w in zip(CASES, fetch, write):
        rd = 2 * f["kib"] * 1024 / REPS / 1e6
        wr = w["kib"] * 1024 / REPS / 1e6
        ar, aw = algorithmic(lay, m, n, k, epi)
        ks = ", ".join(f"{v // REPS}x {kk.split('<')[0].replace('void ', '')}" for kk, v in f["kernels"].items())
        print(f"{name:10s} {lay:6s} {m:7d}x{n:5d}x{k:6d} {epi:5s} {rd:9.1f} {ar / 1e6:9.1f} {rd / (ar / 1e6):5.2f} "
              f"{wr:9.1f} {aw / 1e6:9.1f} {wr / (aw / 1e6):5.2f}  {ks}")
    print("# read MB = 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half); Infinity-Cache hits are included")


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run()

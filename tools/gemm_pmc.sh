#!/bin/bash
# instruction counts of the step's kernels (one PMC pass over the bench command): vector / matrix / scalar / LDS / VMEM
export TMPDIR=/tmp
out=gpurun_out/gemm_pmc; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES -d $out/p1 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt > $out/p1.log 2>&1 || { echo "pass failed"; tail -3 $out/p1.log; }
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/gemm_pmc/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.match(r"_Z\d+([a-z0-9_]+)I?(.{0,28})", k)
        k = (m.group(1) + "<" + m.group(2)) if m else k[:40]
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for k, c in agg.items():
    n = c["SQ_INSTS_VALU"][0]
    v = {x: c[x][1] / max(c[x][0], 1) for x in c}
    rows.append((v.get("SQ_INSTS_VALU", 0) * n, k, n, v))
print(f"{'kernel':58s} {'calls':>6} {'VALU/launch':>12} {'MFMA':>10} {'VALU per MFMA':>13} {'SALU':>10} {'LDS':>9} {'VMEM':>9} {'waves':>7}")
for tot, k, n, v in sorted(rows, reverse=True)[:16]:
    mf = v.get("SQ_INSTS_MFMA", 0)
    print(f"{k:58s} {n:6d} {v.get('SQ_INSTS_VALU',0):12.0f} {mf:10.0f} {(v.get('SQ_INSTS_VALU',0) - mf) / mf if mf else 0:13.2f} {v.get('SQ_INSTS_SALU',0):10.0f} {v.get('SQ_INSTS_LDS',0):9.0f} {v.get('SQ_INSTS_VMEM',0):9.0f} {v.get('SQ_WAVES',0):7.0f}")
PY
rm -rf $out/p1

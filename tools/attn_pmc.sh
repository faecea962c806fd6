#!/bin/bash
# PMC passes over the attention microbenchmark (run on the GPU box): LDS conflicts / activity and instruction-class busy cycles
export TMPDIR=/tmp
out=gpurun_out/attn_pmc; mkdir -p $out
i=0
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/p$i -- python3 tools/attn_bench.py --bwd --dtype fp16 --only fusion --rounds 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/attn_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "fused" if "fused32" in k else "q" if "attn_bwd_q" in k else "kv" if "attn_bwd_kv" in k else "fwd" if "attn_fwd" in k else None
        if k is None: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in ("fwd", "q", "kv", "fused"):
    print(k, {c: round(v[1] / v[0]) for c, v in sorted(agg[k].items())})
PY
rm -rf $out/p*/

#!/bin/bash
# PMC passes over the attention FORWARD microbenchmark for two builds of the library (run on the GPU box):
#   tools/attn_pmc_ab.sh <tag> <lib.so> [attn_bench args]
# instruction counts, active / waiting cycles per class, instruction fetches
export TMPDIR=/tmp
tag=$1; lib=$2; shift 2
out=gpurun_out/attn_pmc_$tag; mkdir -p $out
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  BIFOLD_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/p$i -- python3 tools/attn_bench.py --dtype fp16 --rounds 2 "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        k = k.split("(")[0][:60]
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k, {c: round(v[1] / v[0]) for c, v in sorted(agg[k].items())})
PY
rm -rf $out/p*/

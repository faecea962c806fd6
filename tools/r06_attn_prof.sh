#!/bin/bash
# round 6: per-kernel durations and PMC counters of the attention microbenchmark (forward + backward) on chosen shapes
# usage: tools/r06_attn_prof.sh <tag> "<attn_bench args>"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/r06/prof_$tag; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/attn_bench.py --bwd --dtype fp16 --rounds 2 "$@" > $out/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $out/kt.log; }
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "attn" in r["Name"]:
            print(f'{float(r["AverageNs"])/1e3:10.1f} us  x{r["Calls"]:>4}  {r["Name"][:90]}')
PY
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/p$i -- python3 tools/attn_bench.py --bwd --dtype fp16 --rounds 1 "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        k = k.split("(")[0][:48]
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    c = {n: v[1] / v[0] for n, v in agg[k].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("   wave-life: active %.0f%%  issue-stall %.0f%%  wait %.0f%% | valu-active %.0f%% lds-active %.0f%% | insts: valu %.3g mfma %.3g lds %.3g salu %.3g vmem %.3g | mfma busy cyc %.3g  gui %.3g  lds conflict %.0f%% of lds active" % (
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc,
        100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_MFMA", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_VMEM", 0),
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
rm -rf $out/p*/ $out/kt

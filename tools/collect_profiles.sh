#!/bin/bash
# rocprofv3 evidence for bench.py's default command (run ON the GPU box, from the repo root):
#   tools/collect_profiles.sh <tag>      ->  gpurun_out/prof_<tag>/{stats,fetch,write,mfma}.txt
# Counters are collected in their own passes with --kernel-trace only (no --stats / sys-trace next to --pmc).
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_raw -- $BENCH > $out/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $out/stats.log; exit 1; }
f=$(find $out/stats_raw -name '*kernel_stats.csv' | head -1)
python3 tools/prof_summary.py "$f" 6 > $out/stats.txt
rm -rf $out/stats_raw
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/${name}_raw -- $BENCH > $out/$name.log 2>&1 || { echo "$name pass failed"; tail -5 $out/$name.log; exit 1; }
done
python3 tools/pmc_summary.py $out/fetch_raw $out/write_raw > $out/hbm_traffic.txt
python3 tools/mfma_summary.py $out/mfma_raw > $out/mfma.txt
rm -rf $out/fetch_raw $out/write_raw $out/mfma_raw
head -12 $out/stats.txt; head -12 $out/hbm_traffic.txt; head -14 $out/mfma.txt

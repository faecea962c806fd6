#!/bin/bash
# raw kernel trace of the default bench command -> gpurun_out/trace_<tag>/timeline.txt (tools/trace_timeline.py)
set -o pipefail
tag=${1:-t}
out=gpurun_out/trace_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt > $out/run.log 2>&1 || { echo "trace failed"; tail -5 $out/run.log; exit 1; }
f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py "$f" 3 2 > $out/timeline.txt
head -c 3000000 "$f" > $out/kernel_trace_head.csv
rm -rf $out/raw
head -60 $out/timeline.txt

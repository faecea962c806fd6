#!/bin/bash
# Per-kernel accounting of the step WITHOUT concurrency (side streams folded into the main stream), so that a kernel's
# duration is its own cost: rocprofv3 --kernel-trace --stats of the bench command under BIFOLD_SERIAL_STREAMS=1
# (bifold/models/streams.py).   tools/serial_stats.sh <tag> [extra bench flags]   ->  gpurun_out/prof_<tag>/serial_stats.txt
set -o pipefail
tag=${1:-s}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp BIFOLD_SERIAL_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial_raw -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt "$@" > $out/serial.log 2>&1 || { echo "serial stats pass failed"; tail -5 $out/serial.log; exit 1; }
f=$(find $out/serial_raw -name '*kernel_stats.csv' | head -1)
python3 tools/prof_summary.py "$f" 6 > $out/serial_stats.txt
rm -rf $out/serial_raw
tail -1 $out/serial.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial step under the profiler:', d['ms_per_step'], 'ms', d['dtype'])"
head -30 $out/serial_stats.txt | cut -c1-160

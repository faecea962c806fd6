#!/bin/bash
# cost of bench.py's in-step HIP-event timer: default flags vs --no-kernel-timer, interleaved on one box
set -o pipefail
out=gpurun_out/ab_timer; mkdir -p $out
for rep in 1 2 3; do
  for mode in timer notimer; do
    flags="--steps 10 --warmup 3 --no-cpu-baseline --no-parity"
    [ $mode = notimer ] && flags="$flags --no-kernel-timer"
    timeout -k 10 200 python3 bench.py $flags > $out/${mode}_${rep}.json 2> $out/${mode}_${rep}.err || { echo "$mode failed"; tail -5 $out/${mode}_${rep}.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$out/${mode}_${rep}.json").read().strip().splitlines()[-1])
print("$mode rep=$rep", d["value"], d["ms_per_step"], flush=True)
PY
  done
done

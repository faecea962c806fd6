#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh "VAR=a" "VAR=b" ... (each run twice, interleaved)
set -o pipefail
out=gpurun_out/ab_env
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt"
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for kv in "$@"; do
    i=$((i+1))
    env $kv timeout -k 10 200 $B > $out/b_${i}_$rep.json 2> $out/b_${i}_$rep.err || { echo "setting '$kv' failed"; tail -5 $out/b_${i}_$rep.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$out/b_${i}_$rep.json").read().strip().splitlines()[-1])
print("$kv rep=$rep", d["value"], d["ms_per_step"], flush=True)
PY
  done
done

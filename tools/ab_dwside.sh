#!/bin/bash
# A/B on one box: dW GEMMs of the fusion backward in line (0) vs on a side stream (1: join per layer, 2: join one layer late)
set -o pipefail
out=gpurun_out/ab_dw
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt"
for rep in 1 2; do
  for m in 0 1 2; do
    BIFOLD_DW_SIDE=$m timeout -k 10 200 $B > $out/b_${m}_$rep.json 2> $out/b_${m}_$rep.err || { echo "mode $m failed"; tail -5 $out/b_${m}_$rep.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$out/b_${m}_$rep.json").read().strip().splitlines()[-1])
print("dw_side=$m rep=$rep", d["value"], d["ms_per_step"], flush=True)
PY
  done
done

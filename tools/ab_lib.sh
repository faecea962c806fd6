#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <nameA> <nameB> [reps] ("" = the product library)
set -o pipefail
out=gpurun_out/ab_lib
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt"
reps=${3:-2}
for rep in $(seq 1 $reps); do
  for v in "$1" "$2"; do
    lib=$PWD/bifold_amd/lib/libbifold_hip${v:+_$v}.so
    BIFOLD_HIP_LIB=$lib timeout -k 10 200 $B > $out/b_${v:-new}_$rep.json 2> $out/b_${v:-new}_$rep.err || { echo "variant '$v' failed"; tail -5 $out/b_${v:-new}_$rep.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("$out/b_${v:-new}_$rep.json").read().strip().splitlines()[-1])
print("lib=${v:-new} rep=$rep", d["value"], d["ms_per_step"], flush=True)
PY
  done
done

#!/bin/bash
# tools/ab_tail.sh: the in-launch tail split (BIFOLD_P8_TAIL_MODE=2) against whole tiles (0) at several minimum chunk depths,
# on the shapes whose tile count leaves an under-filled last round (tools/gemm_bench.py, fp16, variant 9, rotated rounds)
out=gpurun_out/ab_tail
mkdir -p $out
run() { # label, env...
  local label=$1; shift
  for only in "vis fc2" "fus fc2" "fus dH" "fus dqkv" "vis out" "fus out"; do
    env "$@" timeout -k 10 120 python3 tools/gemm_bench.py --variants 9 --rounds 7 --dtype fp16 --only "$only" 2>/dev/null | grep -E "^v 9 (vis|fus)" | awk -v l="$label" '{printf "%-22s %s %s %8s us %8s TF/s\n", l, $3, $4, $7, $9}'
  done
}
run "whole tiles" BIFOLD_P8_TAIL_MODE=0
for kt in 12 8 6 4; do run "split, chunks >= $kt" BIFOLD_P8_TAIL_MODE=2 BIFOLD_P8_TAIL_MIN_KT=$kt; done

#!/bin/bash
# A/B on one box: the product library against bifold_amd/lib/libbifold_hip_old.so (attention.hip of the previous commit)
set -o pipefail
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-timer"
for rep in 1 2 3; do
  BIFOLD_ATTN_BWD_WPB=4 BIFOLD_HIP_LIB=$PWD/bifold_amd/lib/libbifold_hip_old.so timeout -k 10 200 $B 2>/dev/null > gpurun_out/ab_old_$rep.json; echo "old $(python3 tools/benchline.py gpurun_out/ab_old_$rep.json)"
  timeout -k 10 200 $B 2>/dev/null > gpurun_out/ab_new_$rep.json; echo "new $(python3 tools/benchline.py gpurun_out/ab_new_$rep.json)"
done

#!/bin/bash
# A/B on one box: the product library against bifold_amd/lib/libbifold_hip_old.so (attention.hip of an earlier commit),
# attention microbenchmark first, then the step
for rep in 1 2; do
  for v in old new; do
    lib=$PWD/bifold_amd/lib/libbifold_hip.so; [ $v = old ] && lib=$PWD/bifold_amd/lib/libbifold_hip_old.so
    echo "$v $(BIFOLD_HIP_LIB=$lib python3 tools/attn_bench.py --bwd --dtype fp16 --only fusion | tr '\n' ' ')"
  done
done
[ "$1" = "step" ] || exit 0
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-timer --no-alt"
for rep in 1 2 3; do
  for v in old new; do
    lib=$PWD/bifold_amd/lib/libbifold_hip.so; [ $v = old ] && lib=$PWD/bifold_amd/lib/libbifold_hip_old.so
    BIFOLD_HIP_LIB=$lib timeout -k 10 200 $B 2>/dev/null > gpurun_out/ab_${v}_$rep.json; echo "$v $(python3 tools/benchline.py gpurun_out/ab_${v}_$rep.json)"
  done
done

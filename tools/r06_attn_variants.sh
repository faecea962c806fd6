#!/bin/bash
# round 6: A/B of library variants (tools/build_variant.sh) on the long-sequence attention shapes, one box
# usage: tools/r06_attn_variants.sh "<variant names, base = product>" [extra attn_bench flags, e.g. --bwd]
set -o pipefail
mkdir -p gpurun_out/r06
SH=${SHAPES:-"--shape fusion128,128,2950,16,48,2 --shape fus224,32,1050,16,48,2 --shape vision384,64,576,12,64,0 --shape vitl,64,577,16,64,0"}
out=gpurun_out/r06/variants_$(date +%H%M%S).txt
for rep in 1 2; do
for v in $1; do
  [ "$v" = "base" ] && lib=$PWD/bifold_amd/lib/libbifold_hip.so || lib=$PWD/bifold_amd/lib/libbifold_hip_$v.so
  echo "== $v (rep $rep)" | tee -a $out
  BIFOLD_HIP_LIB=$lib python tools/attn_bench.py --dtype fp16 $SH $2 2>&1 | grep -v amdgpu.ids | tee -a $out
done
done

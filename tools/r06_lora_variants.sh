#!/bin/bash
# round 6: load-depth variants of csrc/lora.hip (LORA_KU / LORA_UN) on the isolated bench, same box
set -o pipefail
out=gpurun_out/r06; mkdir -p $out
for v in "" _l22 _l88; do
  echo "== libbifold_hip$v.so" | tee -a $out/lora_variants.txt
  BIFOLD_HIP_LIB=$PWD/bifold_amd/lib/libbifold_hip$v.so python3 tools/lora_bench.py 2>&1 | grep -v "amdgpu.ids\|GEMM form" | tee -a $out/lora_variants.txt
done

#!/bin/bash
# round 6: every attention test, then forward + backward timings with the pipelined kernels off / on, one box
set -o pipefail
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -15 > gpurun_out/r06/attn_tests.txt
cat gpurun_out/r06/attn_tests.txt
grep -q "failed\|error" gpurun_out/r06/attn_tests.txt && exit 1
SH=${SHAPES:-"--shape fusion128,128,2950,16,48,2 --shape fus224,32,1050,16,48,2 --shape vision384,64,576,12,64,0 --shape vitl,64,577,16,64,0"}
out=gpurun_out/r06/attn_all_$(date +%H%M%S).txt
for pipe in 0 1 0 1; do
  echo "== BIFOLD_ATTN_PIPE=$pipe" | tee -a $out
  BIFOLD_ATTN_PIPE=$pipe timeout -k 10 300 python tools/attn_bench.py --bwd --dtype fp16 $SH 2>&1 | grep -v amdgpu.ids | tee -a $out
done

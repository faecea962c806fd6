#!/bin/bash
# round 6: the pipelined backward pair -- parity tests, then old / new timings on the long-sequence shapes
set -o pipefail
mkdir -p gpurun_out/r06
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -15 > gpurun_out/r06/bwd_tests.txt
cat gpurun_out/r06/bwd_tests.txt
grep -q "failed\|error" gpurun_out/r06/bwd_tests.txt && exit 1
SH="--shape fusion128,128,2950,16,48,2 --shape fusion16,16,2373,16,48,2 --shape fus224,32,1050,16,48,2 --shape vision384,64,576,12,64,0"
for pipe in 0 1; do
  echo "== BIFOLD_ATTN_PIPE=$pipe" | tee -a gpurun_out/r06/bwd_bench.txt
  BIFOLD_ATTN_PIPE=$pipe timeout -k 10 300 python tools/attn_bench.py --bwd --dtype fp16 $SH 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06/bwd_bench.txt
done

#!/bin/bash
# round 6, GPU call 1: the pipelined forward -- parity tests, then old / new timings on the long-sequence shapes
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention_fwd or masked_segments" 2>&1 | tail -15 > gpurun_out/r06/fwd_tests.txt
cat gpurun_out/r06/fwd_tests.txt
SH="--shape fusion128,128,2950,16,48,2 --shape fusion16,16,2373,16,48,2 --shape fus224,32,1050,16,48,2 --shape vision384,64,576,12,64,0 --shape vitl,64,577,16,64,0"
for pipe in 0 1; do
  echo "== BIFOLD_ATTN_PIPE=$pipe" | tee -a gpurun_out/r06/fwd_bench.txt
  BIFOLD_ATTN_PIPE=$pipe python tools/attn_bench.py --dtype fp16 $SH 2>&1 | tee -a gpurun_out/r06/fwd_bench.txt
done
python tools/attn_bench.py --dtype bf16 $SH 2>&1 | tee -a gpurun_out/r06/fwd_bench.txt

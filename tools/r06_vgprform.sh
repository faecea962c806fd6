#!/bin/bash
# round 6: the pipelined attention kernels built with -mllvm -amdgpu-mfma-vgpr-form against the product library
mkdir -p gpurun_out/r06
BIFOLD_HIP_LIB=$PWD/bifold_amd/lib/libbifold_hip_vgprform.so timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -4
SHAPES="--shape fusion128,128,2950,16,48,2 --shape fus224,32,1050,16,48,2 --shape vitl,64,577,16,64,0" bash tools/r06_attn_variants.sh "base vgprform" --bwd

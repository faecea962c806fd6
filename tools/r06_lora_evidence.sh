#!/bin/bash
# round 6: the LoRA configuration after csrc/lora.hip -- isolated bench, kernel profile (side streams folded in), bench lines
# (unmerged as shipped / GEMM form of the adapter branch / merged), then the whole GPU suite
set -o pipefail
out=gpurun_out/r06; mkdir -p $out
export TMPDIR=/tmp
step() { echo "[$(date +%H:%M:%S)] $*"; }
step "isolated bench"
python3 tools/lora_bench.py 2>&1 | grep -v amdgpu.ids > $out/lora_bench.txt; tail -3 $out/lora_bench.txt
step "kernel profile, LoRA as shipped"
tools/serial_stats.sh r06s_siglip_lora --config siglip_seq --lora --batch 128 > $out/serial_siglip_lora.log 2>&1 && cp gpurun_out/prof_r06s_siglip_lora/serial_stats.txt $out/siglip_seq_lora_b128_kernel_stats.txt || { echo FAILED serial siglip lora; tail -3 $out/serial_siglip_lora.log; }
step "bench lines on this box: riding / up-projection as a pass / GEMM form / merged, two rounds"
bash tools/r06_lora_ab.sh | tee $out/lora_same_box_ab.txt
step "the GPU test suite, observed parity recorded"
rm -f $out/observed_parity.jsonl
BIFOLD_OBSERVED=$out/observed_parity.jsonl timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
step done

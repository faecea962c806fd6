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
step "bench lines: unmerged (fused kernels), unmerged (GEMM form), merged"
timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora.json 2> /dev/null
BIFOLD_LORA_FUSED=0 timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora_gemm_form.json 2> /dev/null
timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --lora-dropout 0 --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora_merged.json 2> /dev/null
for f in bench_line_siglip_seq_lora bench_line_siglip_seq_lora_gemm_form bench_line_siglip_seq_lora_merged; do python3 tools/benchline.py $out/$f.json; done
step "the GPU test suite, observed parity recorded"
rm -f $out/observed_parity.jsonl
BIFOLD_OBSERVED=$out/observed_parity.jsonl timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
step done

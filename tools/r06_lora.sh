#!/bin/bash
# round 6: the LoRA adapter-branch kernels -- parity tests, the isolated bench, then the siglip_seq LoRA bench line
set -o pipefail
out=gpurun_out/r06; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "lora or dropout or low_rank" 2>&1 | tail -15 | tee $out/lora_tests.txt || exit 1
python3 tools/lora_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/lora_bench.txt
timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora_fused.json 2> $out/lora.err || { tail -5 $out/lora.err; exit 1; }
python3 tools/benchline.py $out/bench_line_siglip_seq_lora_fused.json

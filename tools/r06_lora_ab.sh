#!/bin/bash
# round 6: the LoRA configuration on one box -- up-projection riding in the GEMM (default) / as its own pass / GEMM form / merged
set -o pipefail
out=gpurun_out/r06; mkdir -p $out
run() { # name, env, extra flags
  env $2 timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora $3 --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora_$1.json 2> /dev/null
  python3 tools/benchline.py $out/bench_line_siglip_seq_lora_$1.json
}
for rep in 1 2; do
run ride BIFOLD_LORA_FUSED=1 ""
run up_pass BIFOLD_LORA_FUSED=up ""
run gemm_form BIFOLD_LORA_FUSED=0 ""
run merged BIFOLD_LORA_FUSED=1 "--lora-dropout 0"
done

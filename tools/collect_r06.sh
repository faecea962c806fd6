#!/bin/bash
# Round-6 evidence, in three GPU calls (each under gpurun's 20-minute limit; run on the GPU box from the repo root):
#   tools/collect_r06.sh a   config A: serial kernel stats, concurrent stats + PMC passes, attention kernels, timeline, bench line
#   tools/collect_r06.sh b   the other configurations' bench lines + a kernel profile of --config siglip_seq
#   tools/collect_r06.sh c   GEMM against the vendor library, the whole GPU test suite with the observed-parity record
# Everything lands in gpurun_out/r06/; tools/publish_r06.sh copies what is judged into profiles/r06_*.
# Every step writes its own file as it goes (a long silent call is taken for hung).
set -o pipefail
part=${1:-a}
out=gpurun_out/r06
mkdir -p $out
export TMPDIR=/tmp
filt() { grep -v amdgpu.ids; }
step() { echo "[$(date +%H:%M:%S)] $*"; }
case $part in
a)
  step "serial kernel stats fp16"
  tools/serial_stats.sh r06s_fp16 > $out/serial_fp16.log 2>&1 && cp gpurun_out/prof_r06s_fp16/serial_stats.txt $out/serial_kernel_stats_fp16.txt || { echo FAILED serial fp16; tail -3 $out/serial_fp16.log; }
  step "concurrent kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy)"
  tools/collect_profiles.sh r06p > $out/collect.log 2>&1 && cp gpurun_out/prof_r06p/stats.txt $out/bench_kernel_stats.txt && cp gpurun_out/prof_r06p/hbm_traffic.txt $out/hbm_traffic_pmc.txt && cp gpurun_out/prof_r06p/mfma.txt $out/mfma_busy_pmc.txt || { echo FAILED collect_profiles; tail -3 $out/collect.log; }
  step "attention kernels, isolated"
  { timeout -k 5 150 python3 tools/attn_bench.py --bwd --dtype fp16 2>&1 | filt;
    echo "# long sequences (configs[2] at B = 128: 2 950 tokens, query-row mask; the 224-px form; SigLIP / ViT-L towers), BIFOLD_ATTN_PIPE=1 (default) then 0 (rounds 1 - 5's kernels)";
    for pipe in 1 0; do echo "== BIFOLD_ATTN_PIPE=$pipe"; BIFOLD_ATTN_PIPE=$pipe timeout -k 5 200 python3 tools/attn_bench.py --bwd --dtype fp16 --shape fusion128,128,2950,16,48,2 --shape fus224,32,1050,16,48,2 --shape vision384,64,576,12,64,0 --shape vitl,64,577,16,64,0 2>&1 | filt; done; } > $out/attn_bench.txt
  step "LayerNorm passes, isolated (the frozen towers' stream written once per block)"
  timeout -k 5 150 python3 tools/add_ln_bench.py 2>&1 | filt > $out/add_ln_bench.txt
  step "step timeline"
  tools/trace_run.sh r06t > /dev/null 2>&1 && cp gpurun_out/trace_r06t/timeline.txt $out/step_timeline.txt
  step "bench line: default (cpu baseline + alt)"
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err
  python3 tools/benchline.py $out/bench_line.json
  ;;
b)
  step "bench lines: bf16, vitl14, fp8"
  timeout -k 10 300 python3 bench.py --no-alt --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $out/bench_line_bf16.json 2> /dev/null
  timeout -k 10 400 python3 bench.py --no-alt --config vitl14 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line_vitl14.json 2> /dev/null
  timeout -k 10 300 python3 bench.py --no-alt --config fp8 --batch 512 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line_fp8.json 2> /dev/null
  step "  siglip_seq: frozen, LoRA as shipped (lora_dropout 0.01: unmerged), LoRA merged (dropout 0)"
  timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_frozen.json 2> /dev/null
  timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora.json 2> /dev/null
  timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --lora-dropout 0 --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora_merged.json 2> /dev/null
  for f in bench_line_bf16 bench_line_vitl14 bench_line_fp8 bench_line_siglip_seq_frozen bench_line_siglip_seq_lora bench_line_siglip_seq_lora_merged; do python3 tools/benchline.py $out/$f.json; done
  step "kernel profile of --config siglip_seq (frozen towers, B = 128), side streams folded in"
  tools/serial_stats.sh r06s_siglip --config siglip_seq --batch 128 > $out/serial_siglip.log 2>&1 && cp gpurun_out/prof_r06s_siglip/serial_stats.txt $out/siglip_seq_frozen_b128_kernel_stats.txt || { echo FAILED serial siglip; tail -3 $out/serial_siglip.log; }
  step "  ... and with LoRA as shipped"
  tools/serial_stats.sh r06s_siglip_lora --config siglip_seq --lora --batch 128 > $out/serial_siglip_lora.log 2>&1 && cp gpurun_out/prof_r06s_siglip_lora/serial_stats.txt $out/siglip_seq_lora_b128_kernel_stats.txt || { echo FAILED serial siglip lora; tail -3 $out/serial_siglip_lora.log; }
  step "PMC counters of the long-sequence attention kernels (wave life, instruction mix, LDS conflicts)"
  bash tools/r06_attn_prof.sh pipe --shape fusion128,128,2950,16,48,2 --shape vision384,64,576,12,64,0 > $out/attn_pipe_pmc.txt 2>&1 || echo FAILED attn pmc
  ;;
c)
  step "GEMM vs the vendor library, fp16 (same buffers, rotated order)"
  { echo "# tools/gemm_bench.py --variants 9 --blaslt --dtype fp16 --rounds 7: v 9 = the persistent kernel (BK = 64 full-line loop, the default); v99 = hipBLASLt through torch.matmul"; timeout -k 5 400 python3 tools/gemm_bench.py --variants 9 --blaslt --dtype fp16 --rounds 7 2>&1 | filt; } > $out/gemm_vs_hipblaslt_fp16.txt
  step "the GPU test suite, observed parity recorded"
  rm -f $out/observed_parity.jsonl
  BIFOLD_OBSERVED=$out/observed_parity.jsonl timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1
  tail -3 $out/gpu_tests.log
  ;;
esac
step done
ls $out
